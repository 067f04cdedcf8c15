"""On-GPU batch augmentation (csrc/augment.hip through the C ABI) vs golden F11 (the real reference) and the oracle.
Run with `pytest -m gpu`."""
import numpy as np
import pytest
import torch

import amd_pkg
from oracle import augment as oaug
from tests.util import load_golden, hashed_mel

pytestmark = pytest.mark.gpu

pkg = amd_pkg.load()


@pytest.fixture(scope="module")
def paug():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ts_asr_whisper_amd import augment
    return augment


def gpu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_gaussian_noise_bit_exact_vs_reference(paug):
    z = load_golden("f11_augment")
    for i in range(int(z["n_noise"])):
        var, frac, seed = z[f"noise_{i}_cfg"]
        torch.manual_seed(int(seed))
        got = paug.add_gaussian_noise_and_rescale(gpu(z["stno"]), float(var), float(frac)).cpu().numpy()
        assert np.array_equal(got, z[f"noise_{i}_out"]), f"case {i}: max diff {np.abs(got - z[f'noise_{i}_out']).max()}"


def test_soft_segments_vs_reference(paug):
    """Class means are a wave reduction instead of ATen's sum: same dominant class unless two means tie to an ulp, after
    which every element is the same fp32 expression -> bit-exact on the fixture."""
    z = load_golden("f11_augment")
    for i in range(int(z["n_seg"])):
        cp, lo, hi, seed = z[f"seg_{i}_cfg"]
        torch.manual_seed(int(seed))
        got = paug.soft_segment_augmentation(gpu(z["stno"]), float(cp), int(lo), int(hi)).cpu().numpy()
        assert np.array_equal(got, z[f"seg_{i}_out"]), f"case {i}: max diff {np.abs(got - z[f'seg_{i}_out']).max()}"


def test_spec_aug_joint_vs_reference(paug):
    """fp32 tolerance 1e-6 against the reference (bicubic taps; same bound the oracle is pinned with); the zeroed
    mask pattern has to be identical."""
    z = load_golden("f11_augment")
    for i in range(int(z["n_spec"])):
        M, B, Tm, seed = (int(v) for v in z[f"spec_{i}_cfg"])
        mel, stno = hashed_mel(B, M, Tm), z["stno"][:B, :, :Tm // 2]
        torch.manual_seed(seed)
        mo, so = paug.spec_aug_joint(gpu(mel), gpu(stno))
        mo, so = mo.cpu().numpy(), so.cpu().numpy()
        assert np.abs(mo - z[f"spec_{i}_mel_out"]).max() < 1e-6, f"case {i}"
        assert np.abs(so - z[f"spec_{i}_stno_out"]).max() < 1e-6, f"case {i}"
        assert np.array_equal(mo == 0, z[f"spec_{i}_mel_out"] == 0)
        torch.manual_seed(seed)
        omel, ostno = oaug.spec_aug_joint(mel, stno)
        assert np.abs(mo - omel).max() < 5e-7 and np.abs(so - ostno).max() < 5e-7


def test_spec_aug_edge_plans(paug):
    mel, stno = gpu(hashed_mel(2, 80, 64)), torch.softmax(torch.randn(2, 4, 32), 1).cuda()
    # nothing enabled but an empty plan: identity (and a STNO round trip through repeat / mean)
    mo, so = paug.spec_aug_joint(mel, stno, plan=paug.SpecAugPlan())
    assert torch.equal(mo, mel) and torch.equal(so, stno)
    # a "warp" that maps each piece onto itself is the identity as well (integer source positions -> weights 0,1,0,0)
    mo, so = paug.spec_aug_joint(mel, stno, plan=paug.SpecAugPlan(center=20, warped=20))
    assert torch.equal(mo, mel) and torch.equal(so, stno)
    # 80 mel bins: features 80..83 (the STNO rows) are inside [:128] and get masked with the mel (augmentations.py:369)
    plan = paug.SpecAugPlan(fmask=torch.tensor([[[78, 4]], [[0, 0]]], dtype=torch.int32),
                            tmask=torch.tensor([[[10, 6]], [[63, 1]]], dtype=torch.int32))
    mo, so = paug.spec_aug_joint(mel, stno, plan=plan)
    assert float(mo[0, 78:].abs().max()) == 0 and float(so[0, :2].abs().max()) == 0
    assert torch.equal(mo[0, :78, :10], mel[0, :78, :10]) and float(mo[0, :, 10:16].abs().max()) == 0
    assert torch.equal(so[0, 2:, :5], stno[0, 2:, :5]) and float(so[0, :, 5:8].abs().max()) == 0
    assert torch.equal(mo[1, :, :63], mel[1, :, :63]) and float(mo[1, :, 63].abs().max()) == 0
    assert torch.allclose(so[1, :, 31], stno[1, :, 31] / 2) and torch.equal(so[1, :, :31], stno[1, :, :31])
    with pytest.raises(Exception, match="not in-place|does not match"):
        paug.spec_aug_joint(mel, stno[:, :, :16].contiguous())


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_collator_block_vs_oracle_full_size(paug, seed):
    """The whole augmentation block (collators.py:189-214) with the dicow_v3 recipe values at B=16, 30 s inputs."""
    g = torch.Generator().manual_seed(100 + seed)
    B, M, Tn = 16, 128, 1500
    mel = torch.randn(B, M, 2 * Tn, generator=g).clamp_(-1.5, 1.5)
    stno = torch.softmax(torch.randn(B, 4, Tn, generator=g) * 3, dim=1)
    kw = dict(segment_prob=0.9, change_prob=0.1, min_seg_len=5, max_seg_len=50, noise_var=0.2, noise_prob=0.75,
              spec_aug_prob=0.8)
    torch.manual_seed(seed)
    omel, ostno = oaug.augment_batch(mel.numpy(), stno.numpy(), **kw)
    aug = paug.BatchAugmenter(stno_gaussian_noise_var=0.2, stno_gaussian_noise_prob=0.75, stno_segment_augment_prob=0.9,
                              stno_segment_change_prob=0.1, stno_min_segment_length=5, stno_max_segment_length=50,
                              spec_aug_prob=0.8)
    torch.manual_seed(seed)
    batch = {"input_features": mel.cuda(), "stno_mask": stno.cuda()}
    keep = batch["stno_mask"].clone()
    out = aug(batch)
    assert torch.equal(keep, stno.cuda())                                  # the caller's STNO tensor is not modified
    gm, gs = out["input_features"].cpu().numpy(), out["stno_mask"].cpu().numpy()
    assert np.abs(gm - omel).max() < 1e-6 and np.abs(gs - ostno).max() < 1e-6
    assert np.array_equal(gm == 0, omel == 0)


def test_augment_throughput_note(paug):
    """B=64 x 30 s: the three kernels are HBM streams; this just makes sure they run at the bench size."""
    B, M, Tn = 64, 128, 1500
    mel = torch.randn(B, M, 2 * Tn, device="cuda").clamp_(-1.5, 1.5)
    stno = torch.softmax(torch.randn(B, 4, Tn, device="cuda"), 1)
    torch.manual_seed(0)
    aug = paug.BatchAugmenter(stno_gaussian_noise_var=0.2, stno_gaussian_noise_prob=0.75, stno_segment_augment_prob=1.0,
                              spec_aug_prob=1.0)
    out = aug({"input_features": mel, "stno_mask": stno})
    s = out["stno_mask"].sum(1)
    live = s > 0                                                           # 128 mel bins: STNO rows are warped, never masked
    assert bool(live.all()) and float((s - 1).abs().max()) < 1e-3
    frac0 = float((out["input_features"] == 0).float().mean())
    assert 0.02 < frac0 < 0.6


def test_augment_edge_cases(paug):
    """Degenerate plans: no row selected (no draw consumed), no segment changed, every frame in one changed segment, inputs too
    short for the warp and the ratio time mask, a single-row batch."""
    st = torch.softmax(torch.randn(1, 4, 40), 1)
    torch.manual_seed(1)
    a = float(torch.rand(1))
    torch.manual_seed(1)
    out = paug.add_gaussian_noise_and_rescale(st.cuda(), 0.2, 0.75)            # int(1 * 0.75) == 0 rows
    assert torch.equal(out.cpu(), st) and float(torch.rand(1)) == a
    torch.manual_seed(2)
    assert torch.equal(paug.soft_segment_augmentation(st.cuda(), 0.0, 5, 9).cpu(), st)
    torch.manual_seed(3)
    got = paug.soft_segment_augmentation(st.cuda(), 1.0, 40, 40).cpu().numpy()
    torch.manual_seed(3)
    want = oaug.soft_segment_augmentation(st.numpy(), 1.0, 40, 40)
    assert np.array_equal(got, want) and not np.array_equal(got, st.numpy())
    mel = torch.from_numpy(hashed_mel(1, 128, 10))
    st5 = torch.softmax(torch.randn(1, 4, 5), 1)
    torch.manual_seed(4)
    mo, so = paug.spec_aug_joint(mel.cuda(), st5.cuda())
    torch.manual_seed(4)
    omel, ostno = oaug.spec_aug_joint(mel.numpy(), st5.numpy())
    assert np.array_equal(mo.cpu().numpy(), omel) and np.array_equal(so.cpu().numpy(), ostno)    # masks only: exact
    assert torch.equal(so.cpu(), st5)
