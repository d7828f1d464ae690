"""Output stage: the pixel values the product writes equal the reference's two quantisers on the
fixture inputs (tests/golden/output_stage.npz: half-way and out-of-range values included; the PNG quantiser is the PUBLISHED
torchvision.utils.save_image algorithm — torchvision is not installed and the reference pins no version — evaluated with
torch by tests/golden/make_golden.py), the PNG files decode (own decoder AND PIL) to exactly those values, the FID array is
the reference's `samples_N.npz` content (NHWC uint8 `arr_0`), make_npz.py packs a PNG directory into that format, and on the
GPU the device-side quantiser kernel is bit-exact and the whole generate_cifar10.py CLI keeps up with the sampler."""
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


@pytest.mark.gpu
def test_quantize_kernel_bit_exact_and_png_roundtrip(golden_dir, tmp_path):
    """dxmi_quantize_u8 (both modes, both layouts) against the fixture; ImageWriter's PNGs decoded by PIL and by make_npz."""
    from PIL import Image
    from dxmi_hip import ops
    from utils import ImageWriter
    import make_npz
    g = np.load(os.path.join(golden_dir, "output_stage.npz"))
    x = torch.from_numpy(g["x"]).to("cuda:0")
    assert np.array_equal(ops.quantize_u8(x, mode=0, nhwc=True).cpu().numpy(), g["png_restated_hwc"])
    assert np.array_equal(ops.quantize_u8(x, mode=0, nhwc=False).cpu().numpy(), g["png_restated_hwc"].transpose(0, 3, 1, 2))
    assert np.array_equal(ops.quantize_u8(x, mode=1, nhwc=False).cpu().numpy(), g["fid_uint8_nchw"])
    assert np.array_equal(ops.quantize_u8(x, mode=1, nhwc=True).cpu().numpy(), g["fid_uint8_nchw"].transpose(0, 2, 3, 1))
    # a larger random batch against the torch expressions of the generate scripts (same device, fp32)
    gen = torch.Generator(device="cuda:0").manual_seed(4)
    big = torch.randn(64, 3, 32, 32, device="cuda:0", generator=gen) * 0.9
    ref0 = ((big - (-1)) / 2).clamp(0, 1).mul(255).add_(0.5).clamp_(0, 255).to(torch.uint8).permute(0, 2, 3, 1)
    ref1 = ((big + 1) * 127.5).clamp(0, 255).to(torch.uint8)
    assert torch.equal(ops.quantize_u8(big, mode=0), ref0) and torch.equal(ops.quantize_u8(big, mode=1, nhwc=False), ref1)
    # ImageWriter: device quantise -> pinned copy on a side stream -> PNG threads; two batches exercise both host buffers
    d = tmp_path / "generated"
    d.mkdir()
    w = ImageWriter(workers=3)
    w.submit(x, [str(d / f"0_{i}.png") for i in range(6)])
    w.submit(big[:10], [str(d / f"0_{6 + i}.png") for i in range(10)])
    w.submit(x, [str(d / f"1_{i}.png") for i in range(6)])
    w.close()
    for i in range(6):
        assert np.array_equal(np.asarray(Image.open(str(d / f"0_{i}.png"))), g["png_restated_hwc"][i])       # PIL decodes what we wrote
        assert np.array_equal(_decode_png(str(d / f"1_{i}.png")), g["png_restated_hwc"][i])
    assert np.array_equal(np.asarray(Image.open(str(d / "0_15.png"))), ref0[9].cpu().numpy())
    # make_npz (reference README.md:161-164): sorted (rank, index) order, arr_0 uint8 NHWC — 8x8 and 32x32 files cannot be stacked
    d2 = tmp_path / "gen2"
    d2.mkdir()
    w = ImageWriter(workers=2)
    w.submit(big[:12], [str(d2 / f"0_{i}.png") for i in range(12)])
    w.close()
    arr = make_npz.main(["--dir", str(d2), "--out", str(tmp_path / "generated.npz")])
    assert arr.shape == (12, 32, 32, 3) and np.array_equal(np.load(str(tmp_path / "generated.npz"))["arr_0"], ref0[:12].cpu().numpy())


@pytest.mark.gpu
def test_generate_cifar10_cli_keeps_up_with_the_sampler(tmp_path):
    """generate_cifar10.py end to end (20 000 synthetic-weight images as PNG files, --skip_fid) against the bare
    sampler.sample rate measured in the same process tree on the same GPU: the output stage (device quantise, pinned
    double-buffered copy, PNG thread pool) must not cost more than 20 % of the generation rate."""
    import re
    import shutil
    import subprocess
    import sys
    import time
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diffusion-by-maxentirl_amd")
    env = dict(os.environ, LOCAL_RANK="0", WORLD_SIZE="1")
    logdir = str(tmp_path / "run")
    os.makedirs(logdir)
    r = subprocess.run([sys.executable, "generate_cifar10.py", "--log_dir", logdir, "--synthetic", "cifar10_T10", "-n", "20000",
                        "--batchsize", "250", "--skip_fid"], cwd=pkg, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    m = re.search(r"\(([0-9.]+) images/s/rank incl. PNG writing\)", r.stdout)
    assert m, r.stdout[-1000:]
    cli_rate = float(m.group(1))
    files = os.listdir(os.path.join(logdir, "generated"))
    assert len(files) == 20000
    from PIL import Image
    im = np.asarray(Image.open(os.path.join(logdir, "generated", "0_19999.png")))
    assert im.shape == (32, 32, 3) and im.dtype == np.uint8
    shutil.rmtree(logdir, ignore_errors=True)
    # the bare sampler on the same GPU, same batch size
    import configs_builtin
    import dxmi_config
    cfg = configs_builtin.get("cifar10_T10")
    net = dxmi_config.instantiate(cfg.sampler_net)
    sampler = dxmi_config.instantiate(cfg.sampler, net=net).to("cuda:0").eval()
    with torch.no_grad():
        for _ in range(3):
            sampler.sample(250, device="cuda:0")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            sampler.sample(250, device="cuda:0")
        torch.cuda.synchronize()
        bare = 250 * 20 / (time.perf_counter() - t0)
    print(f"generate_cifar10.py: {cli_rate:.0f} images/s incl. PNG files, bare sampler {bare:.0f} images/s")
    assert cli_rate >= 0.8 * bare, (cli_rate, bare)
