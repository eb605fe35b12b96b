"""Same import path as the reference (libs/GANet/functions/GANet.py); implementation: ganet_amd."""
from ganet_amd.functions.GANet import *  # noqa: F401,F403
from ganet_amd.functions.GANet import __all__  # noqa: F401
