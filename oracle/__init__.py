"""ORACLE package: CPU restatement of the reference hot path.  Test infrastructure only -- see oracle/unet3d.py."""
