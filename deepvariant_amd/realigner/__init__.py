"""The window realigner (deepvariant/realigner/): window selection, local assembly and
read -> haplotype -> reference realignment.  Host mirrors of the reference's Python modules
over libdvhip.so; see realigner.py."""
