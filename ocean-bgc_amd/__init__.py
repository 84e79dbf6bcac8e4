"""MI355X-native ocean-BGC column engine: host-side mirror of the reference's
module API (BGC_mod / DMS_mod / MACROS_mod) over the C-ABI library libobgc.so.

The directory name carries a hyphen (project layout); import it as
`ocean_bgc_amd` through tests/conftest.py::load_pkg or __graft_entry__.load_pkg.
"""
__all__ = ["synth"]
