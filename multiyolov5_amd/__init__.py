"""multiyolov5_amd -- the joint detection + segmentation hot path of TomMao23/multiyolov5, rebuilt for MI355X (gfx950).

Layout:  csrc/ (HIP kernels + C ABI, built into lib/libmyolo.so), include/myolo.h (the ABI), engine.py (static launch
plans), models/ and utils/ (host-side mirror of the reference's module / loss / NMS API).  No CPU or ATen fallback.
"""
__version__ = '0.1.0'
