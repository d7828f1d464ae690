"""Host-side mirror of the reference's `models` package for the DxMI hot path.

Same dotted names as the reference (`models.DxMI.var_sampler.VARSampler`, ...), so the reference's
YAML `_target_` entries and the train_*/generate_* scripts resolve to these classes unchanged.
All compute goes through dxmi_hip (libdxmi_hip.so); there is no CPU implementation here.
"""
