"""ganet_amd -- GA-Net's guided-aggregation hot path (SGA, LGA, GetCostVolume,
DisparityRegression) as hand-written HIP kernels for AMD Instinct MI355X (gfx950),
behind the reference's own operator API (libs/GANet/{functions,modules}/GANet.py).
See DESIGN.md and INTEGRATION.md."""
__version__ = "0.1.0"
