"""FID evaluation path of the DxMI scripts, the part that needs no Inception weights (reference: pytorch_fid/)."""
