"""Self-consistency known-answer tests for oracle/sched_ref.py (the scheduler arithmetic lives in the absent
dependency diffusers, so it is "parity unpinned"; these pin the published algorithm's invariants)."""
import torch

from oracle import sched_ref as S


def test_ddim_constants_and_timesteps():
    s = S.DDIMSchedulerRef()
    assert float(s.init_noise_sigma) == 1.0
    ac = s.alphas_cumprod
    assert abs(float(ac[0]) - (1 - 0.00085)) < 1e-7
    assert 0.0046 < float(ac[-1]) < 0.0047  # SD scaled-linear schedule: alpha_bar_999 ~= 0.00466
    s.set_timesteps(50)
    assert s.timesteps.tolist() == list(range(980, -1, -20))
    s.set_timesteps(1000)
    assert s.timesteps[0] == 999 and s.timesteps[999] == 0
    x = torch.randn(2, 4, 8, 8)
    assert torch.equal(s.scale_model_input(x, 10), x)


def test_ddim_step_inverts_add_noise_with_exact_eps():
    s = S.DDIMSchedulerRef()
    s.set_timesteps(50)
    x0, eps = torch.randn(2, 4, 8, 8), torch.randn(2, 4, 8, 8)
    t = int(s.timesteps[10])
    xt = s.add_noise(x0, eps, torch.tensor([t]))
    out = s.step(eps, t, xt)
    torch.testing.assert_close(out.pred_original_sample, x0, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out.prev_sample, s.add_noise(x0, eps, torch.tensor([t - 20])), rtol=1e-4, atol=1e-4)
    last = s.step(eps, 0, s.add_noise(x0, eps, torch.tensor([0])))  # prev_t < 0 -> final_alpha_cumprod = 1
    torch.testing.assert_close(last.prev_sample, x0, rtol=1e-4, atol=1e-4)


def test_euler_a_sigmas_and_step():
    s = S.EulerAncestralSchedulerRef()
    assert abs(float(s.init_noise_sigma) - 14.6146) < 1e-3  # sigma_max of the SD schedule
    s.set_timesteps(12)
    assert len(s.sigmas) == 13 and float(s.sigmas[-1]) == 0.0
    assert float(s.init_noise_sigma) == float(s.sigmas[0]) == float(s.sigmas.max())
    assert abs(float(s.timesteps[0]) - 999.0) < 1e-4 and float(s.timesteps[-1]) == 0.0
    x = torch.randn(1, 4, 8, 8) * s.init_noise_sigma
    t = s.timesteps[0]
    xin = s.scale_model_input(x, t)
    torch.testing.assert_close(xin, x / (s.sigmas[0] ** 2 + 1) ** 0.5)
    # with eps = x / sigma the predicted x0 is 0; the ancestral step then returns pure fresh noise * sigma_up
    eps = x / s.sigmas[0]
    torch.manual_seed(0)
    out = s.step(eps, t, x)
    torch.manual_seed(0)
    noise = torch.randn(eps.shape)
    sf, st = s.sigmas[0], s.sigmas[1]
    sigma_up = (st ** 2 * (sf ** 2 - st ** 2) / sf ** 2) ** 0.5
    sigma_down = (st ** 2 - sigma_up ** 2) ** 0.5
    torch.testing.assert_close(out.prev_sample, x * (sigma_down / sf) + noise * sigma_up, rtol=1e-5, atol=1e-5)
    # variance preserving split: sigma_down^2 + sigma_up^2 == sigma_to^2
    assert abs(float(sigma_down ** 2 + sigma_up ** 2 - st ** 2)) < 1e-4
    s.set_timesteps(1000)
    assert float(s.timesteps[500]) == 499.0
