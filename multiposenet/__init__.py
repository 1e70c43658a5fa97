"""multiposenet — MI355X-native MultiPoseNet hot path (see multiposenet.pytorch_amd)."""
