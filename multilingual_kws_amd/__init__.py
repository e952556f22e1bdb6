"""multilingual_kws_amd -- the MI355X (gfx950) hot path of harvard-edge/multilingual_kws.

micro-frontend features -> EfficientNet-B0 embedding -> few-shot head, as hand-written HIP kernels in
lib/libmkws_hip.so behind the C-ABI of include/mkws.h, with the reference's own Python surface
(multilingual_kws_amd.embedding.input_data / transfer_learning) on top.  There is no CPU fallback:
every compute entry point raises if the HIP library or a gfx950 device is missing.
"""
__version__ = "0.1.0"
