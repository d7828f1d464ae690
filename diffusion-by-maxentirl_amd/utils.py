"""Small helpers of the CLI surface (reference utils.py:21, :140, :251-273)."""
import errno
import os
from collections import OrderedDict

import torch


def get_rank():
    return torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0


def print0(*args, **kwargs):
    if get_rank() == 0:
        print(*args, **kwargs)


def mkdir_p(path):
    try:
        os.makedirs(path)
    except OSError as e:
        if not (e.errno == errno.EEXIST and os.path.isdir(path)):
            raise


def remove_module(d):
    return OrderedDict((k[len("module."):], v) for k, v in d.items())


def fix_legacy_dict(d):
    """Unwrap {'model': ...} / {'state_dict': ...} checkpoints and strip a DDP 'module.' prefix
    (reference utils.py:263-273) — needed to load the pretrained DDPM checkpoint."""
    keys = list(d.keys())
    if "model" in keys:
        d = d["model"]
    if "state_dict" in keys:
        d = d["state_dict"]
    keys = list(d.keys())
    if len(keys) > 1 and "module." in keys[1]:
        d = remove_module(d)
    return d


def weight_norm(net):
    """sqrt(sum ||p||^2) over parameters (reference utils.py:140-147)."""
    total = 0.0
    for p in net.parameters():
        total += p.detach().float().pow(2).sum().item()
    return total ** 0.5


# ------------------------------------------------------------------------------------------ output stage
def to_uint8_nhwc(x01):
    """[N,3,H,W] float in [0,1] (any device) -> uint8 [N,H,W,3] on the host, torchvision.utils.save_image rounding
    (x*255 + 0.5, clamp, truncate; generate_cifar10.py:205-209) — quantised on the device, ONE device->host copy."""
    return x01.mul(255).add_(0.5).clamp_(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous().cpu().numpy()


def _png_bytes(img_hwc_u8, level):
    import struct
    import zlib
    h, w, c = img_hwc_u8.shape
    assert c == 3 and img_hwc_u8.dtype.name == "uint8"
    raw = b"".join(b"\x00" + img_hwc_u8[y].tobytes() for y in range(h))     # filter type 0 on every scanline

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
            + chunk(b"IDAT", zlib.compress(raw, level)) + chunk(b"IEND", b""))


def write_png_batch(batch_nhwc_u8, paths, workers=8, level=3, pool=None):
    """Write a batch of 8-bit RGB PNGs from a thread pool (zlib releases the GIL).  The per-image PIL path costs ~1 ms of
    Python per 32x32 image - more than the GPU needs to generate it.
    pool: a caller-owned ThreadPoolExecutor -> returns the futures without waiting, so encoding batch i overlaps the
    generation of batch i+1."""
    from concurrent.futures import ThreadPoolExecutor

    def one(i):
        with open(paths[i], "wb") as f:
            f.write(_png_bytes(batch_nhwc_u8[i], level))

    if pool is not None:
        return [pool.submit(one, i) for i in range(len(paths))]
    if workers <= 1 or len(paths) < 4:
        for i in range(len(paths)):
            one(i)
        return []
    with ThreadPoolExecutor(max_workers=workers) as ex:
        list(ex.map(one, range(len(paths))))
    return []


class ImageWriter:
    """Output stage of the generate scripts without stalling the sampler: the batch is quantised on the device by
    dxmi_quantize_u8 (NCHW fp32 -> NHWC uint8, the reference's rounding), copied to one of two PINNED host buffers on a side
    stream (the next batch's kernels keep the compute stream busy meanwhile), and encoded / written by a thread pool once the
    copy's event has fired.  Replaces `sample.cpu()` + a per-image `save_image` loop (generate_cifar10.py:203-209,
    generate_large.py:36-41: ~1 ms of Python per 32x32 image, more than the GPU needs to generate it)."""

    def __init__(self, workers=8, level=3):
        from concurrent.futures import ThreadPoolExecutor
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self.level = level
        self.stream = torch.cuda.Stream()
        self.slots = [None, None]          # (pinned buffer, device tensor kept alive, event, futures)
        self.i = 0
        self.pending = []

    def submit(self, sample_nchw, paths, mode=0):
        """sample_nchw: fp32 [N,3,H,W] on the device (sampler output in [-1, 1]); paths: one file name per image."""
        from dxmi_hip import ops
        u8 = ops.quantize_u8(sample_nchw.contiguous().float(), mode=mode, nhwc=True)
        k = self.i & 1
        self.i += 1
        slot = self.slots[k]
        if slot is not None:
            for f in slot[3]:               # the buffer is free once its previous batch is on disk
                f.result()
        if slot is None or slot[0].shape != u8.shape:
            host = torch.empty(u8.shape, dtype=torch.uint8, pin_memory=True)
        else:
            host = slot[0]
        ev = torch.cuda.Event()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            host.copy_(u8, non_blocking=True)
            ev.record()
        arr, level = host.numpy(), self.level
        nper = max(1, (len(paths) + 7) // 8)

        def run(lo, hi):
            ev.synchronize()
            for j in range(lo, hi):
                with open(paths[j], "wb") as f:
                    f.write(_png_bytes(arr[j], level))

        futs = [self.pool.submit(run, lo, min(lo + nper, len(paths))) for lo in range(0, len(paths), nper)]
        self.slots[k] = (host, u8, ev, futs)
        self.pending += futs
        return futs

    def close(self):
        for f in self.pending:
            f.result()
        self.pending = []
        self.pool.shutdown()
