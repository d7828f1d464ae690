"""Output stage: the pixel values the product writes equal the reference's two quantisers on the
fixture inputs (tests/golden/output_stage.npz: half-way and out-of-range values included), the PNG files decode to
exactly those values, and the FID array is the reference's `samples_N.npz` content (NHWC uint8 `arr_0`)."""
import os
import struct
import zlib

import numpy as np
import pytest
import torch


def _decode_png(path):
    """Minimal decoder for the 8-bit RGB, filter-0 PNGs utils.write_png_batch emits (and any PNG using filter type 0)."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I", data[pos:pos + 4])[0], data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == (zlib.crc32(tag + body) & 0xFFFFFFFF)
        if tag == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            assert (depth, ctype) == (8, 2)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 3)


def _check(device, golden_dir, tmp_path):
    from generate_cifar10 import rescale
    from utils import to_uint8_nhwc, write_png_batch
    g = np.load(os.path.join(golden_dir, "output_stage.npz"))
    x = torch.from_numpy(g["x"]).to(device)
    # generate_cifar10.py:205-209 and generate_large.py:36-41: rescale -> clamp -> save_image
    u8 = to_uint8_nhwc(rescale(x).clamp(0, 1))
    assert u8.dtype == np.uint8 and np.array_equal(u8, g["png_restated_hwc"])
    u8b = to_uint8_nhwc(((x + 1) / 2).clamp(0, 1))
    assert np.array_equal(u8b, g["png_restated_hwc"])
    paths = [str(tmp_path / f"0_{i}.png") for i in range(len(u8))]
    write_png_batch(u8, paths, workers=2)
    for i, pth in enumerate(paths):
        assert np.array_equal(_decode_png(pth), g["png_restated_hwc"][i])
    # generate_large.py:43: the FID / samples_N.npz quantiser
    fid = ((x + 1) * 127.5).clamp(0, 255).to(torch.uint8)
    assert np.array_equal(fid.cpu().numpy(), g["fid_uint8_nchw"])
    np.savez(str(tmp_path / "samples_6.npz"), fid.permute(0, 2, 3, 1).cpu().numpy())
    arr = np.load(str(tmp_path / "samples_6.npz"))["arr_0"]
    assert arr.shape == (6, 8, 8, 3) and np.array_equal(arr, g["fid_uint8_nchw"].transpose(0, 2, 3, 1))
    # the two quantisers differ exactly where the reference's do (x*255+0.5 truncation vs (x+1)*127.5 truncation)
    assert (u8.transpose(0, 3, 1, 2) != g["fid_uint8_nchw"]).any()


def test_output_stage_values_cpu(golden_dir, tmp_path):
    _check("cpu", golden_dir, tmp_path)


@pytest.mark.gpu
def test_output_stage_values_gpu(golden_dir, tmp_path):
    _check("cuda:0", golden_dir, tmp_path)
