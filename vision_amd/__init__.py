"""vision_amd — MI355X-native box geometry (IoU, NMS, YOLO decode, RoI pooling, anchor k-means)
behind the Python call signatures of crapine1119/vision's hot path.  See DESIGN.md."""

__all__ = ["ops", "yolo_v1", "yolo_v2", "rcnn_utils", "fast_rcnn", "cluster", "shard"]
