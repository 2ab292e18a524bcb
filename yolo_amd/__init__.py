"""MI355X-native YOLOv3 inference hot path (Darknet-53 forward, head decode,
confidence-filtered per-class NMS) behind the Python surface of the
jayden199012/yolo reference.

Submodules
  yolo_amd.utilis   drop-in for the reference's utilis.py
  yolo_amd.yolo_v3  drop-in for the reference's yolo_v3.py
  yolo_amd.cfg      generator for the cfg text / params dict the reference expects
  yolo_amd.engine   fused execution plan (descriptors for libyolo_amd.so)
  yolo_amd.dist     batch sharding and the detection gather for multi-GPU runs

Importing yolo_amd.utilis / yolo_amd.yolo_v3 loads libyolo_amd.so and fails
loudly when it has not been built.
"""
__all__ = ["cfg"]
