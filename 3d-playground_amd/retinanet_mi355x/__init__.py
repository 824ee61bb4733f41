"""retinanet_mi355x -- MI355X-native (gfx950) 3D-RetinaNet hot path behind the reference's Python surface.

Layout:
  _hip.py     ctypes binding of libretinanet_mi355x.so (C ABI: include/retinanet_mi355x.h)
  ops.py      functional ops (anchors, IoU/assign, focal loss with backward, decode, post-process, homography)
  arch.py     static network description (reference state_dict keys)
  synth.py    seeded synthetic frames / labels / weights
The drop-in packages next to this one (``retinanet/``, ``flat2d/retinanet/``, ``homography.py``) re-export the
reference's module and class names on top of these.
"""
__version__ = "0.1"
