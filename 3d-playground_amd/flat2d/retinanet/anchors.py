"""Same names as the reference's R/anchors.py."""
from retinanet_mi355x.modules import Anchors  # noqa: F401
