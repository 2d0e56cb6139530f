"""siammot-mi355x: MI355X-native (gfx950) EMM tracker head for SiamMOT.

Host-side mirror of the reference's operator interface for the EMM hot path
(reference siammot/modelling/track_head/EMM/) over a C-ABI HIP library
(``csrc/libsmot_emm.so``, declared in ``include/smot_emm.h``).
"""
__version__ = "0.1.0"
