"""MI355X-native vectorised RWARE step engine (drop-in for the step path of
semitable/robotic-warehouse: reset / step / FLATTENED observation of `rware.warehouse.Warehouse`).

The directory is named `robotic-warehouse_amd` (not an identifier); import it as `rware_amd`
(the shim module at the repository root) or via importlib.
"""
from .enums import DEFAULT_IMAGE_LAYERS, Action, Direction, ImageLayer, ObservationType, RewardType
from .layout import Layout, layout_from_params, layout_from_str, obs_length
from .registry import Pipeline, all_ids, capture_pipelines, env_kwargs, make_pipelines, make_vec, register_gymnasium, shard_seeds, streams_overlap
from .vector_env import STATE_FIELDS, WarehouseVecEnv

__all__ = [
    "Action", "Direction", "ImageLayer", "DEFAULT_IMAGE_LAYERS", "ObservationType", "RewardType", "Layout", "layout_from_params",
    "layout_from_str", "obs_length", "all_ids", "env_kwargs", "make_vec", "make_pipelines", "capture_pipelines", "Pipeline", "streams_overlap", "register_gymnasium", "shard_seeds",
    "WarehouseVecEnv", "STATE_FIELDS",
]
__version__ = "0.1.0"
