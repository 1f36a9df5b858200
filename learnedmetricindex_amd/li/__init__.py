"""Mirror of the reference's `li` package (/root/reference/search/li): same module, class and
function names, same signatures, MI355X arithmetic underneath.

Use it either as `learnedmetricindex_amd.li` or, like the reference (which puts `search/` on
sys.path, setup.py:8-10), by putting `learnedmetricindex_amd/` on sys.path and importing `li`.
"""
