"""alonet — MI355X-native drop-in for the dense-vision hot path of aloception's ``alonet`` package.

Only the sub-packages on (or right next to) the hot path exist here: ``alonet.deformable_detr`` (multi-scale
deformable attention behind the reference's operator API) and ``alonet.raft`` (all-pairs correlation behind the
``corr_block=`` hook).  The native code lives in ``libalo_hotpath.so`` (``aloception-oss_amd/csrc``), bound through
``alo_hip``.
"""
import os

ALONET_ROOT = os.path.dirname(os.path.abspath(__file__))
