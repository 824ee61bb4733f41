"""Same names as the reference's D/anchors.py."""
from retinanet_mi355x.modules import Anchors  # noqa: F401
