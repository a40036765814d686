"""sceneego_amd — MI355X-native implementation of SceneEgo's depth-aware voxel pose hot path.

Public surface mirrors the reference modules for that path:
  sceneego_amd.voxel_net_depth.VoxelNetwork_depth   <- network/voxel_net_depth.py
  sceneego_amd.v2v.V2VModel                         <- network/v2v.py
  sceneego_amd.pose_resnet.get_pose_net             <- network/pose_resnet.py
  sceneego_amd.op                                   <- utils/op.py (+ init-time geometry)
  sceneego_amd.fisheye.FishEyeCameraCalibrated      <- utils/fisheye/FishEyeCalibrated.py
  sceneego_amd.config.load_config                   <- utils/cfg.py
The compute kernels are in sceneego_amd/csrc (HIP, gfx950) behind the C ABI of include/sceneego_hip.h.
"""
from .config import EasyDict, load_config  # noqa: F401

__all__ = ["EasyDict", "load_config", "VoxelNetwork_depth", "VoxelNetDepth"]


def __getattr__(name):
    if name in ("VoxelNetwork_depth", "VoxelNetDepth"):
        from .voxel_net_depth import VoxelNetwork_depth
        return VoxelNetwork_depth
    raise AttributeError(name)
