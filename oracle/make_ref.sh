#!/bin/bash
# TEST / MEASUREMENT INFRASTRUCTURE — stages the UNMODIFIED reference package into the git-ignored oracle/_ref/.
#
# The reference (semitable/robotic-warehouse) is pure Python: "building" it means making the two files of its step path
# importable — rware/__init__.py (registry, rware/__init__.py:7-39) and rware/warehouse.py (Warehouse.step, :804-946).
# They are copied byte for byte from where they lie under /root/reference into oracle/_ref/rware/ (never into git history:
# oracle/_ref/ is in .gitignore, but not in .gpurunignore, so the staged copy travels to the GPU box like a built .so).
# There oracle/ref_runner.py finds it when /root/reference is absent, and bench.py's `cpu_baseline` leg can time the
# reference's own pure-Python step on the GPU box's host cores (BASELINE.json north_star) — kind "reference".
# A manifest with the sha256 of every staged file is written beside them; tests/test_oracle_vs_reference.py checks the
# staged copy against /root/reference where both exist.
# Called by __graft_entry__.build() when /root/reference exists; a no-op (exit 0) otherwise.
set -eu
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="${RWARE_REFERENCE_ROOT:-/root/reference}"
DST="$HERE/_ref"
if [ ! -f "$SRC/rware/warehouse.py" ]; then
  echo "make_ref: no reference tree at $SRC — nothing staged (an existing $DST is left as it is)"
  exit 0
fi
rm -rf "$DST"
mkdir -p "$DST/rware"
cp "$SRC/rware/__init__.py" "$SRC/rware/warehouse.py" "$DST/rware/"
[ -f "$SRC/LICENSE" ] && cp "$SRC/LICENSE" "$DST/LICENSE"
(cd "$DST" && sha256sum rware/__init__.py rware/warehouse.py > MANIFEST.sha256)
echo "make_ref: staged $(wc -l < "$DST/MANIFEST.sha256") files of the unmodified reference into $DST"
