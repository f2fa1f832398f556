"""Randomised whole-frame parity: seeded random small shapes and parameters (group / chunk / segment remainders, short and
long arms, with and without the scanline stage), every output against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 23])
def test_random_frames_vs_oracle(gpu_ready, orc, seed):
    import torch
    from stm_amd import device_api as dev, synth
    rng = np.random.RandomState(seed)
    for case in range(10):
        H = int(rng.randint(3, 90)); W = int(rng.randint(3, 400)); D = int(rng.choice([3, 8, 16, 17, 31, 64, 65, 100]))
        zd = int(rng.randint(0, D)); usd = int(rng.choice([1, 5, 17, 34, 40, 63])); lsd = int(rng.randint(1, usd + 1))
        hslo = bool(rng.randint(0, 2))
        sbs, _ = synth.sbs_frame(H, W, D, zd, seed=1000 * seed + case)
        p = dev.FrameParams(num_disp=D, zero_disp=zd, usd=usd, lsd=lsd)
        dl = torch.zeros(H, W, dtype=torch.float32, device="cuda")
        dr = torch.zeros_like(dl)
        out = torch.zeros(H, W, 3, dtype=torch.uint8, device="cuda")
        dev.d_adcensus_stm(torch.from_numpy(sbs).cuda(), dl, dr, out, p, stages=3 | (0x100 if hslo else 0))
        torch.cuda.synchronize()
        want = orc.adcensus_stm(sbs, H, W, p.num_views, p.angle, D, zd, p.ad_coeff, p.census_coeff, p.ucd, p.lcd, usd, lsd,
                                p.thresh_s, p.thresh_h, hslo=hslo)
        where = (seed, case, H, W, D, zd, usd, lsd, hslo)
        assert np.array_equal(dl.cpu().numpy(), want["disp_l"]), where
        assert np.array_equal(dr.cpu().numpy(), want["disp_r"]), where
        assert np.array_equal(out.cpu().numpy(), want["interlaced"]), where
