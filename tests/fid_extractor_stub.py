"""A weight-free stand-in for the FID feature extractor in CLI tests (`--fid_extractor fid_extractor_stub:PatchFeatures`):
16 deterministic features per image with the reference extractor's call convention (`model(batch)[0]` -> [B, dims, 1, 1],
pytorch_fid/fid_score.py:208).  Not a quality metric: it exists so the FID FLOW of the generate scripts runs end to end."""
import torch


class PatchFeatures(torch.nn.Module):
    dims = 16

    def forward(self, x):                      # x [B, 3, H, W] in [0, 1]
        q = torch.nn.functional.adaptive_avg_pool2d(x.float(), 2).flatten(1)                         # 12: channel x quadrant means
        q2 = torch.nn.functional.adaptive_avg_pool2d(x[:, :1].float() ** 2, 2).flatten(1)            # 4: second moments of channel 0
        return [torch.cat([q, q2], 1)[:, :, None, None]]
