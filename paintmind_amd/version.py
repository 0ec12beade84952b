"""Package version; the C ABI has its own number (pmhip_abi_version in include/pmhip.h)."""
__version__ = "0.1.0"
