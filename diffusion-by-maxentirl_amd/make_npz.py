"""Pack a directory of generated PNGs into the `.npz` batch the ADM / TensorFlow evaluator reads (reference README.md:161-164:
`python make_npz.py --dir <log_dir>/generated --out <log_dir>/generated.npz`; evaluations/evaluator.py:138-139 opens `arr_0`,
uint8 [N, H, W, 3]).  The reference's README names this script but its snapshot does not contain it; this is the documented
behaviour.  Files are decoded by a thread pool (PIL when installed, else the filter-0 decoder of the PNGs utils.write_png_batch
emits) in sorted (rank, index) order."""
import argparse
import os
import re
import struct
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def _decode_png_plain(path):
    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, w, h = 8, b"", None, None
    while pos < len(data):
        n, tag = struct.unpack(">I", data[pos:pos + 4])[0], data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            if (depth, ctype, interlace) != (8, 2, 0):
                raise ValueError(f"{path}: only 8-bit RGB non-interlaced PNGs are supported without PIL")
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    if (raw[:, 0] != 0).any():
        raise ValueError(f"{path}: filtered scanlines need PIL")
    return raw[:, 1:].reshape(h, w, 3)


def read_png(path):
    try:
        from PIL import Image
    except ImportError:
        return _decode_png_plain(path)
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


def _key(name):
    nums = [int(t) for t in re.findall(r"\d+", name)]
    return (nums, name)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", required=True, help="directory of PNG files")
    ap.add_argument("--out", required=True, help="output .npz path")
    ap.add_argument("--workers", type=int, default=8)
    a = ap.parse_args(argv)
    files = sorted((f for f in os.listdir(a.dir) if f.lower().endswith(".png")), key=_key)
    if not files:
        raise SystemExit(f"no PNG files under {a.dir}")
    with ThreadPoolExecutor(max_workers=a.workers) as ex:
        imgs = list(ex.map(lambda f: read_png(os.path.join(a.dir, f)), files))
    arr = np.stack(imgs).astype(np.uint8)
    np.savez(a.out, arr)                      # -> arr_0, uint8 [N, H, W, 3]
    print(f"wrote {a.out}: {arr.shape} uint8 from {len(files)} files")
    return arr


if __name__ == "__main__":
    main()
