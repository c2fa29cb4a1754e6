"""Shared by tests/test_hip_train.py and tests/ddp_train_worker.py: a reduced-width replica of the whole Stage-1 stack."""
import torch

CFG = dict(in_channels=4, model_channels=64, out_channels=4, num_res_blocks=2, attention_resolutions=[4, 2, 1],
           channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True, transformer_depth=1, context_dim=64, legacy=False)


def fixed_face_detector(image_np, T=20):
    """Stands where the reference runs RetinaFace on a decoded image: one face, a box of 5/8 x 5/8 of the frame (39 % of the area:
    past the reference's 'too-large' bound of 36 %, so the face-suppression branch is live too), confidence 0.995."""
    H, W = image_np.shape[:2]
    return [(W * 0.25, H * 0.1875, W * 0.625, H * 0.625, 0.995)]


def trainer_setup(dev, accum=1, process_group=None, ffn_lora=False, embedding_manager=False, stage2=False, faces=False):
    """Reduced-width replica of the whole Stage-1 stack: CLIP encoders hidden 128 / 3 layers, U-Nets model_channels 64."""
    from adaface_dev_amd import rng
    from adaface_dev_amd.adaface.arc2face_models import CLIPTextModelWrapper, clip_text_config
    from adaface_dev_amd.adaface.face_id_to_ada_prompt import Arc2Face_ID2AdaPrompt
    from adaface_dev_amd.adaface.unet_teachers import Arc2FaceTeacher
    from adaface_dev_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from adaface_dev_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    from adaface_dev_amd.ldm.trainer import DistillTrainer, LossScaler
    ccfg = clip_text_config(hidden_size=128, num_attention_heads=2, num_hidden_layers=3, intermediate_size=512)
    ucfg = dict(CFG, context_dim=128)
    ld = LatentDiffusion(ucfg)
    rng.load_synth_weights(ld.model.diffusion_model, seed=41)
    teacher_unet = UNetModel(**ucfg)
    rng.load_synth_weights(teacher_unet, seed=42)
    id2ada = Arc2Face_ID2AdaPrompt(clip_config=ccfg)
    rng.load_synth_weights(id2ada.text_to_image_prompt_encoder, seed=43)
    rng.load_synth_weights(id2ada.subj_basis_generator.prompt2token_proj, seed=44)
    text_enc = CLIPTextModelWrapper(ccfg)
    rng.load_synth_weights(text_enc, seed=45)
    sds = dict(student=ld.model.diffusion_model.state_dict(), teacher=teacher_unet.state_dict(),
               arc2face=id2ada.text_to_image_prompt_encoder.state_dict(),
               sbg=id2ada.subj_basis_generator.prompt2token_proj.state_dict(), text=text_enc.state_dict())
    sds = {k: {n: v.detach().clone() for n, v in sd.items()} for k, sd in sds.items()}
    if embedding_manager:           # the reference's conditioning path: hooked text encoder + EmbeddingManager around the same modules
        from adaface_dev_amd.ldm.modules.encoders.modules import FrozenCLIPEmbedder
        ld.instantiate_cond_stage(FrozenCLIPEmbedder(transformer=text_enc, clip_config=ccfg, last_layers_skip_weights=[1, 1]))
        ld.instantiate_embedding_manager({"id2ada_prompt_encoder": id2ada})
    ld = ld.to(dev)
    ld.unet_teacher = Arc2FaceTeacher(teacher_unet.to(dev))
    if ffn_lora:
        lora = ld.model.set_up_ffn_loras(lora_rank=16, lora_dropout=0.0)
        with torch.no_grad():
            for n, p in lora.named_parameters():
                if "lora_B" in n:
                    p.copy_(rng.synth_input(n, p.shape, seed=82, scale=0.3))
                elif "lora_A" in n:       # (the module's own init draws from the global generator: two setups would differ)
                    p.copy_(rng.synth_input(n, p.shape, seed=82, scale=p[0].numel() ** -0.5))
    if stage2:
        # Stage 2 (compositional distillation): a priming U-Net with classifier-free guidance, the unconditional prompt embedding,
        # trainable attention DoRA adapters on the captured layers, and the comp_distill FFN adapters
        from adaface_dev_amd.adaface.unet_teachers import UNetTeacher
        priming_unet = UNetModel(**ucfg)
        rng.load_synth_weights(priming_unet, seed=46)
        ld.comp_distill_priming_unet = UNetTeacher(priming_unet.to(dev), cfg_scale_range=(2, 4), p_uses_cfg=1.0, name="comp_priming")
        ld.uncond_context = (rng.synth_input("s2.uncond", (1, 77, 128), seed=46).to(dev), [""], {})
        for p in ld.model.diffusion_model.parameters():
            p.requires_grad_(False)
        alora = ld.model.set_up_attn_loras(lora_rank=16, lora_dropout=0.0)
        with torch.no_grad():
            for n, p in alora.named_parameters():
                if "lora_B" in n:
                    p.copy_(rng.synth_input(n, p.shape, seed=84, scale=0.2))
                elif "lora_A" in n:
                    p.copy_(rng.synth_input(n, p.shape, seed=84, scale=p[0].numel() ** -0.5))
    if stage2 or faces:
        # what the face-gated loss terms need: a first-stage decoder (reduced width) and the ArcFace wrapper around a detector; the
        # detector here "finds" one fixed face box in every decoded image (RetinaFace itself is an external package)
        from adaface_dev_amd.evaluation.arcface_resnet import resnet_face18
        from adaface_dev_amd.ldm.modules.arcface_wrapper import ArcFaceWrapper, FaceCropper
        vae = ld.instantiate_first_stage(dict(ch=32, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, attn_resolutions=[], dropout=0.0, in_channels=3,
                                              resolution=128, z_channels=4, double_z=True))
        with torch.no_grad():
            for n, p in vae.named_parameters():
                p.copy_(rng.synth_tensor(n, p.shape, seed=90))
        net = resnet_face18()
        rng.load_synth_weights(net, seed=47)
        ld.arcface = ArcFaceWrapper(net.to(dev).eval(), FaceCropper(fixed_face_detector))
        ld = ld.to(dev)
    tr = DistillTrainer(ld, id2ada.to(dev), text_enc.to(dev), accumulate_grad_batches=accum, warm_up_steps=0,
                        loss_scaler=LossScaler(init_scale=2.0 ** 10), process_group=process_group, stage=2 if stage2 else 1)
    return tr, sds, ucfg
