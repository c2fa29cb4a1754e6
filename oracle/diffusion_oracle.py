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


def _collate(dicts):
    """ldm/util.py:1112-1126."""
    out = {}
    for k, v in dicts[0].items():
        col = [d[k] for d in dicts]
        out[k] = sum(col, []) if isinstance(v, list) else torch.cat(col, dim=0) if torch.is_tensor(v) else _collate(col)
    return out


def _split(d, n):
    """ldm/util.py:1129-1165 (reverse of _collate)."""
    res = [{} for _ in range(n)]
    for k, v in d.items():
        if isinstance(v, list):
            per = len(v) // n
            parts = [v[i * per:(i + 1) * per] for i in range(n)]
        elif torch.is_tensor(v):
            parts = list(torch.split(v, v.size(0) // n, dim=0))
        else:
            parts = _split(v, n)
        for i in range(n):
            res[i][k] = parts[i]
    return res


def _detach(o):
    if torch.is_tensor(o):
        return o.detach()
    if isinstance(o, dict):
        return {k: _detach(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(_detach(v) for v in o)
    return o


def guided_denoise_full(model, tables, x_start, noise, t, cond_context, default_uncond_context, uncond_emb=None, img_mask=None,
                        subj_indices=None, normalize_cross_attn=False, mix_sc_mc_attn=False, batch_part_has_grad="all",
                        do_pixel_recon=False, cfg_scale=-1, capture_ca_activations=False, res_hidden_states_gradscale=1,
                        use_attn_lora=False, use_ffn_lora=False, ffn_lora_adapter_name=None, ffn_coin=None):
    """LatentDiffusion.guided_denoise with every gradient mode, ddpm.py:1597-1750, over the wrapper protocol
    ``model(x, t, (prompt_emb, prompts, extra_info)) -> eps`` (which leaves ``extra_info['ca_layers_activations']``).
    PINNED by tests/golden/guided_denoise.npz (the reference method itself, driven around a stand-in wrapper).

    'subject-compos' (:1635-1707): the batch is [SS, SC, SR(=sc_comp_rep), MC] blocks of one instance each.  SS and SR run without
    gradient (SR with the caller's normalize_cross_attn), then either SC and MC jointly as one batch with
    ``mix_attn_mats_in_batch`` (no attention LoRA; MC's eps and activations detached) or SC alone with gradient and MC alone
    without gradient and without any LoRA.  The FFN LoRA is kept on only when a coin ``torch.rand(1) < 0.5`` says so (:1638;
    ``ffn_coin`` overrides the draw)."""
    import copy

    def apply(x, tt, ctx, attn_lora, ffn_lora, name):
        ctx[2]["use_attn_lora"], ctx[2]["use_ffn_lora"], ctx[2]["ffn_lora_adapter_name"] = attn_lora, ffn_lora, name
        return model(x, tt, ctx)

    def sliced(x, tt, ctx, idx, grad, attn_lora, ffn_lora, name):
        emb, prompts, extra = ctx
        with torch.set_grad_enabled(grad):
            return apply(x[idx], tt[idx], (emb[idx], [prompts[i] for i in idx], extra), attn_lora, ffn_lora, name)

    x_noisy = q_sample(tables, x_start, t, noise)
    extra = cond_context[2]
    extra.update(capture_ca_activations=capture_ca_activations, res_hidden_states_gradscale=res_hidden_states_gradscale,
                 img_mask=img_mask, normalize_cross_attn=normalize_cross_attn, subj_indices=subj_indices)
    acts = None
    if batch_part_has_grad in ("none", "all"):
        with torch.set_grad_enabled(batch_part_has_grad == "all" and torch.is_grad_enabled()):
            eps = apply(x_noisy, t, cond_context, use_attn_lora, use_ffn_lora, ffn_lora_adapter_name)
        if capture_ca_activations:
            acts = extra["ca_layers_activations"]
    elif batch_part_has_grad == "subject-compos":
        coin = bool(torch.rand(1) < 0.5) if ffn_coin is None else ffn_coin
        use_ffn_lora = use_ffn_lora and coin

        def ctx_with(**kw):
            e = copy.copy(extra)
            e.update(kw)
            return (cond_context[0], cond_context[1], e)
        c_ss = ctx_with(normalize_cross_attn=False, mix_attn_mats_in_batch=False)
        e_ss = sliced(x_noisy, t, c_ss, [0], False, use_attn_lora, use_ffn_lora, ffn_lora_adapter_name)
        c_sr = ctx_with(normalize_cross_attn=normalize_cross_attn, mix_attn_mats_in_batch=False)
        e_sr = sliced(x_noisy, t, c_sr, [2], False, use_attn_lora, use_ffn_lora, ffn_lora_adapter_name)
        if mix_sc_mc_attn:
            c_sm = ctx_with(normalize_cross_attn=False, mix_attn_mats_in_batch=True)
            e_sm = sliced(x_noisy, t, c_sm, [1, 3], True, False, use_ffn_lora, ffn_lora_adapter_name)
            e_sc, e_mc = e_sm.chunk(2, dim=0)
            e_mc = e_mc.detach()
            a_sc, a_mc = _split(c_sm[2]["ca_layers_activations"], 2)
            a_mc = _detach(a_mc)
        else:
            c_sc = ctx_with(normalize_cross_attn=normalize_cross_attn, mix_attn_mats_in_batch=False)
            e_sc = sliced(x_noisy, t, c_sc, [1], True, use_attn_lora, use_ffn_lora, ffn_lora_adapter_name)
            a_sc = c_sc[2]["ca_layers_activations"]
            c_mc = ctx_with(normalize_cross_attn=False)
            e_mc = sliced(x_noisy, t, c_mc, [3], False, False, False, ffn_lora_adapter_name)
            a_mc = c_mc[2]["ca_layers_activations"]
        eps = torch.cat([e_ss, e_sc, e_sr, e_mc], dim=0)
        if capture_ca_activations:
            acts = _collate([c_ss[2]["ca_layers_activations"], a_sc, c_sr[2]["ca_layers_activations"], a_mc])
    else:
        raise ValueError(batch_part_has_grad)
    if cfg_scale > 1:
        if uncond_emb is None:
            uncond_emb = default_uncond_context[0].repeat(x_noisy.shape[0], 1, 1)
        un = (uncond_emb, default_uncond_context[1] * x_noisy.shape[0], copy.copy(default_uncond_context[2]))
        with torch.no_grad():
            e_un = apply(x_noisy, t, un, False, use_ffn_lora, ffn_lora_adapter_name)
        eps = eps * cfg_scale - e_un * (cfg_scale - 1)
    x_recon = predict_start_from_noise(tables, x_noisy, t, eps) if do_pixel_recon else None
    return eps, x_recon, acts
