"""learnedmetricindex_amd -- MI355X (gfx950) implementation of the LearnedMetricIndex query hot
path (MLP leaf prediction -> top-n_buckets -> exact inner-product bucket scan -> top-k merge)
behind the reference's own `li` Python API.  See DESIGN.md / INTEGRATION.md.

`learnedmetricindex_amd.li` mirrors `/root/reference/search/li` (same module and class names);
`learnedmetricindex_amd._capi` is the ctypes binding of liblmi_hip.so (include/lmi_hip.h).
"""
__version__ = "0.1.0"
