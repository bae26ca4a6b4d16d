"""N > 1 path on CPU: world_size-2 and -3 `gloo` runs of the frame gather (the path's one
exchange step).  Each rank fills ONLY its own row bands of a local framebuffer with the
bytes a render would produce (taken from a reference frame), then gather_frame() must
reassemble the exact full frame on rank 0 -- RGB and all three YUV420 planes."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, band, out_path):
    sys.path.insert(0, ROOT)
    pt = importlib.import_module(PKG)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = W * H
    rng = np.random.default_rng(42)                      # same reference frame on every rank
    ref_rgb = torch.from_numpy(rng.integers(0, 256, total * 3, dtype=np.uint8))
    ref_yuv = torch.from_numpy(rng.integers(0, 256, total + 2 * (total // 4), dtype=np.uint8))
    rgb = torch.zeros_like(ref_rgb)
    yuv = torch.zeros_like(ref_yuv)
    ranges = pt.band_row_ranges(H, band, world, rank)
    # "render": copy only this rank's bands (pack from the reference, unpack locally)
    mine = pt.pack_bands(ref_rgb, ref_yuv, W, H, ranges, torch)
    pt.unpack_bands(mine, rgb, yuv, W, H, ranges)
    packed = pt.pack_bands(rgb, yuv, W, H, ranges, torch)
    pt.gather_frame(packed, W, H, band, rgb, yuv)
    if rank == 0:
        ok = bool(torch.equal(rgb, ref_rgb) and torch.equal(yuv, ref_yuv))
        open(out_path, "w").write("ok" if ok else "mismatch")
    else:
        # non-root ranks must still hold only their own bands
        other = [r for r in range(H) if not any(a <= r < b for a, b in ranges)]
        assert all(int(rgb[r * W * 3:(r + 1) * W * 3].sum()) == 0 for r in other[:5])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H,band", [(2, 64, 36, 8), (3, 48, 50, 4), (2, 32, 6, 8),
                                              (8, 1920, 1080, 8)])     # the scaling run: 135 bands, ranks own 17 or 16
def test_gather_frame_gloo(tmp_path, world, W, H, band):
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(world, _free_port(), W, H, band, str(out)), nprocs=world, join=True)
    assert out.read_text() == "ok"


def test_single_rank_gather_is_identity(pt):
    W, H, band = 16, 8, 2
    total = W * H
    rgb = torch.arange(total * 3, dtype=torch.int64).to(torch.uint8)
    yuv = torch.arange(total + 2 * (total // 4), dtype=torch.int64).to(torch.uint8)
    r2, y2 = rgb.clone(), yuv.clone()
    packed = pt.pack_bands(rgb, yuv, W, H, pt.band_row_ranges(H, band, 1, 0), torch)
    pt.gather_frame(packed, W, H, band, r2, y2)
    assert torch.equal(r2, rgb) and torch.equal(y2, yuv)
