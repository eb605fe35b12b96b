"""Full-model harness: the reference's OWN models (models/GANet_deep.py, models/GANet11.py) as the
caller of this repository's drop-in ops (`libs/` import paths -> ganet_amd -> libganet_hip.so).  Nothing of the model
is re-implemented here: `refmodel` locates the reference's model code, `steps` holds the training / inference step
around it (loss mix, DDP, SyncBatchNorm, checkpoint keys as in train.py / predict.py), `fuse` swaps in the opt-in fused
op chains of ganet_amd.modules.fused."""
