"""TDAC calibration-set generators — the five entry points of the reference's scripts/calibration.py
(`TDAC_cifar_calib_data_generator` :12, `TDAC_bedroom_…` :156, `TDAC_church_…` :263, `TDAC_imagenet_…` :371,
`TDAC_coco_…` :502): one recipe (FP trajectories, features of the middle attention block at every step, density +
cosine scores -> samples per step, random step assignment) over four samplers; they return tuples."""
import torch

from ddim.functions.denoising import cali_generalized_steps
from qdiff.utils import AttentionMap
from edadm.tdac import tdac_allocate, shuffled_step_list, pick_by_step


def TDAC_cifar_calib_data_generator(model, config, lamda, calib_num_samples, num_samples, device, diffusion,
                                    class_cond=True):
    """-> (calib_x[N,3,H,W], t[N] (+ cls)): one FP trajectory batch of `num_samples`, features = input of
    mid.attn_1 at every step, density radius 3.0."""
    seq = diffusion.seq
    shape = (num_samples, 3, config.data.image_size, config.data.image_size)
    img = torch.randn(*shape, device=device)
    hook = AttentionMap(model.mid.attn_1)
    feature_map, all_sample = [], None
    for now_rt, sample_t in enumerate(cali_generalized_steps(model=model, seq=seq, x=img, b=diffusion.betas,
                                                             eta=diffusion.args.eta, args=diffusion.args)):
        if now_rt == 0:
            continue
        feature_map.append(hook.feature[0])
        if now_rt == len(seq):
            all_sample = sample_t[:-1]
            break
    hook.remove()
    _, _, _, t_num = tdac_allocate(feature_map, lamda, calib_num_samples, 3.0)
    t = shuffled_step_list(t_num, device)
    parts = [pick_by_step(all_sample, t[i * num_samples:(i + 1) * num_samples])
             for i in range(int(calib_num_samples / num_samples))]
    calib_data = torch.cat(parts).to(device)
    calib_t = torch.tensor([seq[(len(seq) - 1) - int(s)] for s in t]).to(device)
    if class_cond:
        return calib_data, calib_t, torch.tensor([1] * num_samples, device=device).long()
    return calib_data, calib_t


def TDAC_imagenet_calib_data_generator(model, args, calib_num_samples, num_samples, device, num_timesteps):
    """-> (calib_x, t, index, cond, uncond) for class-conditional LDM with classifier-free guidance (scripts/calibration.py:371-499).

    With several ranks (edadm.dist.world) the trajectory batches -- independent DDIM runs, the whole cost of this stage -- are
    sharded: rank r runs the contiguous block edadm.dist.shard_batches gives it, picks its rows of the calibration set locally and
    ONE all_gather_into_tensor per returned tensor completes them everywhere ("TDAC trajectory generation shards by
    sample").  The start noises are drawn up front by every rank, one torch.randn per batch in batch order, as THIS repository's
    sampler draws them (eda-dm_amd/ldm/models/diffusion/ddim.py: the start noise per sample() call, and no randn_like in a step
    whose sigma is 0), the step allocation and the permutation come from rank 0: the tuple is bit-identical to THIS build's
    one-rank tuple.  (The reference's p_sample_ddim calls noise_like in every step whatever sigma is, ddim_control.py:250, so its
    generator stream for batches i > 0 differs from both: the one-rank tuple of this build was never the reference's stream for
    i > 0 either -- the fixtures pin batch 0 and the allocation.)  eta > 0 keeps the replicated form (per-step noise interleaves
    with the start noises).  calib_num_samples must be a multiple of num_samples (checked on every rank before any collective)."""
    from ldm.models.diffusion.ddim_control import DDIMSampler_control
    from edadm import dist as edist
    import torch.distributed as tdist
    uc = None
    if args.scale != 1.0:
        uc = model.get_learned_conditioning({model.cond_stage_key: torch.tensor(calib_num_samples * [1000]).to(model.device)})
    c = model.get_learned_conditioning({model.cond_stage_key: args.data[:calib_num_samples].to(model.device)})
    shape = list(getattr(args, "latent_shape", [3, 64, 64]))
    sampler = DDIMSampler_control(model)
    unet = model.model.diffusion_model
    hook = AttentionMap(getattr(unet, "model", unet).middle_block[1])
    nb = int(calib_num_samples / num_samples)
    rank, world = edist.world()
    sharded = world > 1 and float(args.ddim_eta) == 0.0 and nb >= world and (world - 1) * ((nb + world - 1) // world) < nb
    if world > 1 and nb * num_samples != calib_num_samples:
        # every rank sees the same arguments, so every rank raises here -- before the broadcast below could strand the others
        raise ValueError("calib_num_samples (%d) must be a multiple of num_samples (%d)" % (calib_num_samples, num_samples))
    mine = edist.shard_batches(nb) if sharded else list(range(nb))
    x_T = [torch.randn([num_samples] + shape, device=device) for _ in range(nb)] if sharded else [None] * nb
    samples, feature_map, ts = {}, None, None
    with torch.no_grad():
        for i in mine:
            sl = slice(i * num_samples, (i + 1) * num_samples)
            out = sampler.sample(S=args.custom_steps, conditioning=c[sl], batch_size=num_samples, shape=shape,
                                 verbose=False, unconditional_guidance_scale=args.scale,
                                 unconditional_conditioning=None if uc is None else uc[sl], eta=args.ddim_eta,
                                 x_T=x_T[i], hooks=[hook] if i == 0 else None)
            intermediates = out[1]
            if i == 0:
                feature_map = out[2]
            samples[i] = intermediates['x_inter'][:-1]
            ts = intermediates['ts']
    hook.remove()
    cond = c[:nb * num_samples]                       # intermediates['cond'][0] of batch i is c[sl] (ddim.py:151-154)
    uncond = uc[:nb * num_samples] if uc is not None else None
    if sharded:
        # rank 0 owns trajectory batch 0, whose mid-block features score the steps: allocation + permutation travel from there
        t = torch.empty(nb * num_samples, dtype=torch.long, device=device)
        if rank == 0:
            _, _, _, t_num = tdac_allocate(feature_map, args.lamda, calib_num_samples, 3.0)
            t.copy_(shuffled_step_list(t_num, device))
        else:
            torch.randperm(nb * num_samples)          # the draw rank 0 makes in shuffled_step_list: the generators stay in step
        if tdist.get_backend() == "gloo" and t.is_cuda:
            h = t.cpu()
            tdist.broadcast(h, src=0)
            t.copy_(h)
        else:
            tdist.broadcast(t, src=0)
        local = {i: pick_by_step(samples[i], t[i * num_samples:(i + 1) * num_samples]) for i in mine}
        calib_data = edist.gather_rows(local, nb)
    else:
        all_samples = [torch.cat([samples[i][k] for i in range(nb)]) for k in range(num_timesteps)]
        _, _, _, t_num = tdac_allocate(feature_map, args.lamda, calib_num_samples, 3.0)
        t = shuffled_step_list(t_num, device)
        calib_data = pick_by_step(all_samples, t)
    index = (num_timesteps - 1) - t
    calib_t = torch.stack([ts[int(s)][0] for s in t]).to(device)
    return calib_data, calib_t, index, cond, uncond


def _tdac_unconditional_ldm(model, args, calib_num_samples, num_samples, device, num_timesteps, fixup_ge):
    """bedroom / church (:156-262, :263-370): unconditional LDM, DDIMSampler, density radius 0.3."""
    from ldm.models.diffusion.ddim import DDIMSampler
    unet = model.model.diffusion_model
    shape = [unet.in_channels, unet.image_size, unet.image_size]
    ddim = DDIMSampler(model)
    hook = AttentionMap(getattr(unet, "model", unet).middle_block[1])
    samples, feature_map, ts = [], None, None
    with torch.no_grad():
        for i in range(int(calib_num_samples / num_samples)):
            out = ddim.sample(args.custom_steps, batch_size=num_samples, shape=shape, eta=args.eta, verbose=False,
                              hooks=[hook] if i == 0 else None)
            if i == 0:
                feature_map = out[2]
            samples.append(out[1]['x_inter'][:-1])
            ts = out[1]['ts']
    hook.remove()
    all_samples = [torch.cat([s[k].to(device) for s in samples]) for k in range(num_timesteps)]
    _, _, _, t_num = tdac_allocate(feature_map, args.lamda, calib_num_samples, 0.3, fixup_ge=fixup_ge)
    t = shuffled_step_list(t_num, device)
    index = (num_timesteps - 1) - t
    calib_data = pick_by_step(all_samples, t)
    calib_t = torch.stack([ts[int(s)][0] for s in t]).to(device)
    return calib_data, calib_t, index


def TDAC_bedroom_calib_data_generator(model, args, calib_num_samples, num_samples, device, num_timesteps):
    """-> (calib_x, t, index)  (scripts/calibration.py:156-262)."""
    return _tdac_unconditional_ldm(model, args, calib_num_samples, num_samples, device, num_timesteps, fixup_ge=False)


def TDAC_church_calib_data_generator(model, args, calib_num_samples, num_samples, device, num_timesteps):
    """-> (calib_x, t, index); differs from bedroom only in the `>= 0` fix-up of the rounding remainder (:332)."""
    return _tdac_unconditional_ldm(model, args, calib_num_samples, num_samples, device, num_timesteps, fixup_ge=True)


def TDAC_coco_calib_data_generator(model, args, calib_num_samples, num_samples, device, num_timesteps):
    """-> (calib_x, t, index, cond, uncond, t_next) for text-conditional LDM (Stable Diffusion), PLMS or DDIM
    (scripts/calibration.py:502-638); density radius 0.3."""
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.models.diffusion.plms import PLMSSampler
    uc = model.get_learned_conditioning(calib_num_samples * [""]) if args.scale != 1.0 else None
    c = model.get_learned_conditioning(args.list_prompts[:calib_num_samples])
    shape = [args.C, args.H // args.f, args.W // args.f]
    sampler = PLMSSampler(model) if args.plms else DDIMSampler(model)
    unet = model.model.diffusion_model
    hook = AttentionMap(getattr(unet, "model", unet).middle_block[1])
    samples, cond, uncond, feature_map, ts, ts_next = [], [], [], None, None, None
    with torch.no_grad():
        for i in range(int(calib_num_samples / num_samples)):
            sl = slice(i * num_samples, (i + 1) * num_samples)
            out = sampler.sample(S=args.custom_steps, conditioning=c[sl], batch_size=num_samples, shape=shape,
                                 verbose=False, unconditional_guidance_scale=args.scale,
                                 unconditional_conditioning=None if uc is None else uc[sl], eta=args.ddim_eta,
                                 x_T=None, hooks=[hook] if i == 0 else None)
            inter = out[1]
            if i == 0:
                feature_map = out[2]
            samples.append(inter['x_inter'][:-1])
            ts = inter['ts']
            ts_next = inter.get('ts_next', inter['ts'][1:] + inter['ts'][-1:])
            cond.append(inter['cond'][0].to(device))
            if uc is not None:
                uncond.append(inter['uncond'][0].to(device))
    hook.remove()
    all_samples = [torch.cat([s[k].to(device) for s in samples]) for k in range(num_timesteps)]
    _, _, _, t_num = tdac_allocate(feature_map, args.lamda, calib_num_samples, 0.3)
    t = shuffled_step_list(t_num, device)
    index = (num_timesteps - 1) - t
    calib_data = pick_by_step(all_samples, t)
    calib_t = torch.stack([ts[int(s)][0] for s in t]).to(device)
    calib_t_next = torch.stack([ts_next[int(s)][0] for s in t]).to(device)
    return calib_data, calib_t, index, torch.cat(cond), (torch.cat(uncond) if uncond else None), calib_t_next
