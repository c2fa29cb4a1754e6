"""CPU oracle for the diffusion wrappers around the U-Net.  TEST INFRASTRUCTURE ONLY.

numpy/torch-CPU restatement of the noise schedule, q_sample, the classifier-free-guidance
combine and the DDIM update.  Imported only by tests/, smoke() and bench.py's cpu_baseline.

Parity status: PINNED against (a) the in-code known answers of the reference
(DDIM-50 timesteps ddim.py:29-35 and the alpha-bar table ddim.py:263-271) and (b) tables
produced by the reference's own ``make_beta_schedule`` / ``make_ddim_timesteps`` /
``make_ddim_sampling_parameters`` and a reference ``p_sample_ddim`` run, committed in
tests/golden/schedule.npz and tests/golden/ddim_step.npz by tests/golden/gen_golden.py.
"""
import numpy as np
import torch


def make_beta_schedule_linear(n_timestep=1000, linear_start=0.00085, linear_end=0.012):
    """ldm/modules/diffusionmodules/util.py:21-25 ("linear" = linspace of sqrt, squared), fp64."""
    return (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2).numpy()


def register_schedule(betas):
    """DDPM.register_schedule, ldm/models/diffusion/ddpm.py:294-345.  fp64 math, fp32 tables."""
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    return {
        "betas": f32(betas),
        "alphas_cumprod": f32(ac),
        "alphas_cumprod_prev": f32(ac_prev),
        "sqrt_alphas_cumprod": f32(np.sqrt(ac)),
        "sqrt_one_minus_alphas_cumprod": f32(np.sqrt(1.0 - ac)),
        "sqrt_recip_alphas_cumprod": f32(np.sqrt(1.0 / ac)),
        "sqrt_recipm1_alphas_cumprod": f32(np.sqrt(1.0 / ac - 1)),
        "posterior_variance": f32(post_var),
        "posterior_mean_coef1": f32(betas * np.sqrt(ac_prev) / (1.0 - ac)),
        "posterior_mean_coef2": f32((1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac)),
    }


def make_ddim_timesteps(num_ddim_timesteps, num_ddpm_timesteps=1000):
    """util.py:46-60, 'uniform'."""
    c = num_ddpm_timesteps // num_ddim_timesteps
    return np.asarray(list(range(0, num_ddpm_timesteps, c))) + 1


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta=0.0):
    """util.py:63-77.  alphacums: fp32 table."""
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return sigmas, alphas, alphas_prev


def q_sample(tables, x_start, t, noise):
    """ddpm.py:395-398 + extract_into_tensor util.py:99-102."""
    a = torch.from_numpy(tables["sqrt_alphas_cumprod"])[t].reshape(-1, 1, 1, 1)
    s = torch.from_numpy(tables["sqrt_one_minus_alphas_cumprod"])[t].reshape(-1, 1, 1, 1)
    return a * x_start + s * noise


def predict_start_from_noise(tables, x_t, t, noise):
    """ddpm.py:389-393."""
    a = torch.from_numpy(tables["sqrt_recip_alphas_cumprod"])[t].reshape(-1, 1, 1, 1)
    s = torch.from_numpy(tables["sqrt_recipm1_alphas_cumprod"])[t].reshape(-1, 1, 1, 1)
    return a * x_t - s * noise


def cfg_combine(e_cond, e_uncond, guidance_scale):
    """ddim.py:255: e = e_u + g (e_c - e_u)."""
    return e_uncond + guidance_scale * (e_cond - e_uncond)


def ddim_update(x, e_t, a_t, a_prev, sigma_t=0.0):
    """ddim.py:279-302 with sigma = 0 noise term dropped (eta = 0).  Scalars are fp32 like
    torch.full((b,1,1,1), table[index]) in the reference."""
    a_t = torch.tensor(a_t, dtype=torch.float32)
    a_prev = torch.tensor(a_prev, dtype=torch.float32)
    # np.sqrt(1. - ddim_alphas) on the fp32 table (ddim.py:62): fp32 in, fp32 out
    sqrt_one_minus_at = torch.tensor(np.sqrt(np.float32(1.0) - np.float32(a_t.item())), dtype=torch.float32)
    pred_x0 = (x - sqrt_one_minus_at * e_t) / a_t.sqrt()
    dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e_t
    x_prev = a_prev.sqrt() * pred_x0 + dir_xt
    return x_prev, pred_x0


def guide_scale_sequence(total_steps, guidance_scale):
    """Linear guidance annealing, ddim.py:166-181, 216-219.  Returns the scale used at each step."""
    if isinstance(guidance_scale, (list, tuple)):
        max_g, min_g = guidance_scale
    else:
        min_g = max_g = max(2.0, guidance_scale)
    delta = (max_g - min_g) / (total_steps - 1)
    out, g = [], max_g
    for i in range(total_steps):
        out.append(g)
        g = g - delta if i <= total_steps - 1 else 1
    return out


def guided_denoise(eps_fn, tables, x_start, noise, t, cond_ctx, uncond_ctx=None, cfg_scale=-1, do_pixel_recon=False, uncond_eps_fn=None):
    """LatentDiffusion.guided_denoise, ddpm.py:1597-1750 (batch_part_has_grad 'all' / 'none' are the same arithmetic): q_sample ->
    eps_c = U-Net(x_t, t, cond) -> if cfg_scale > 1: eps = eps_c * s - eps_u * (s - 1) (:1741) with eps_u from the unconditional
    context, whose extra_info is a copy of ``uncond_context[2]`` -- i.e. WITHOUT the conditional pass's img_mask / flags (:1728-1735),
    hence the separate ``uncond_eps_fn`` -> optional x0 = predict_start_from_noise (:1747).  eps_fn(x, t, ctx) is the U-Net."""
    x_noisy = q_sample(tables, x_start, t, noise)
    eps = eps_fn(x_noisy, t, cond_ctx)
    if cfg_scale > 1:
        eps = eps * cfg_scale - (uncond_eps_fn or eps_fn)(x_noisy, t, uncond_ctx).detach() * (cfg_scale - 1)
    x_recon = predict_start_from_noise(tables, x_noisy, t, eps) if do_pixel_recon else None
    return eps, x_recon
