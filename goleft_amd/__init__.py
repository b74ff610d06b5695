"""goleft_amd -- MI355X-native per-base depth engine behind `goleft depth`.

Only the hot path of brentp/goleft's `depth` subcommand lives here
(/root/reference/depth/depth.go): HIP kernels + C ABI in csrc/, and the
host-side mirror of the reference's `depth` front end (flags, tiling,
BED formatting) in depth.py.  There is no CPU fallback: importing the
engine without the built HIP library raises.
"""
__version__ = "0.1.0"
