from fastgen_amd.networks.EDM.network import EDMPrecond  # noqa: F401
