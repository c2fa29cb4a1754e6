"""Host-side mirror of the reference's ``ldm/modules/arcface_wrapper.py`` (``MaskedGrad`` :9-27, ``ArcFaceWrapper.embed_image_tensor``
:89-166, ``calc_arcface_align_loss`` :171-240) and of the in-tree half of ``evaluation/retinaface_pytorch.py`` (``RetinaFaceClient.crop_faces``
:150-245): the face crop -> grey -> 128 x 128 -> ResNetFace-18 embedding -> cosine alignment / suppression losses that the Stage-2 and
recon iterations read (``ddpm.py:2511-2535``).

What is NOT in the reference tree is the detector network itself (``retinaface.pre_trained_models.get_model("biubug6")``, an external
package with its own weights): ``FaceCropper`` therefore takes a ``detect_faces(image_uint8_hwc, T) -> [(x, y, w, h, confidence), ...]``
callable from the caller -- a real detector, boxes known to the data pipeline, or ``no_faces`` -- and does everything the reference's
client does around it (largest face per instance, clipping, minimum size T, crops resized on the INPUT tensor so the graph is kept,
background faces flattened, full-frame box + mask 0 for instances without a face).

The embedding network is ``evaluation/arcface_resnet.py`` of this package (HIP kernels, frozen, fp16) with its input-gradient node
(``FaceEncodeFn``); the decoded image comes from the VAE decoder's node (``VAEDecodeFn``), so the alignment / suppression losses
back-propagate into the x0 prediction exactly as in the reference: decode -> crop -> bilinear resize -> grey -> ``MaskedGrad`` ->
ResNetFace-18 -> cosine / squared-embedding losses.  An embedding module without a backward may declare ``inference_only = True``; it
is then refused for a tensor that needs its gradient rather than silently detached.  With ``no_faces`` (synthetic data) every
face-gated term is exactly zero, as in the reference."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class MaskedGrad(torch.autograd.Function):
    """Identity whose gradient is multiplied by a fixed mask (arcface_wrapper.py:9-27)."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return x

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask, None


def gen_masked_grad_layer(mask=None):
    return (lambda x: x) if mask is None else (lambda x: MaskedGrad.apply(x, mask.to(x.device)))


def no_faces(image_np, T=20):
    """The detector of a pipeline without one: never finds a face (every face-gated loss term is then exactly zero)."""
    return []


def image_to_uint8(image_ts):
    """[3, H, W] in about [-1, 1] -> uint8 [H, W, 3] (retinaface_pytorch.py:9-22)."""
    a = np.clip(image_ts.detach().float().cpu().numpy().transpose(1, 2, 0), -1, 1)
    return ((a + 1) * 127.5).astype(np.uint8)


def images_to_uint8(images_ts):
    """``image_to_uint8`` of a whole batch [B, 3, H, W] -> uint8 [B, H, W, 3] with ONE device-to-host copy: clip, scale and the truncating
    cast run on the device (same fp32 arithmetic), so a Stage-2 micro-batch hands its 16 decoded images to the detector after one
    synchronisation instead of sixteen 3 MB float copies."""
    u8 = ((images_ts.detach().float().clamp(-1, 1) + 1) * 127.5).to(torch.uint8)
    return u8.permute(0, 2, 3, 1).contiguous().cpu().numpy()


def to_device_async(t, device):
    """A small HOST tensor onto ``device`` without stopping the host: through pinned memory and a non-blocking copy (a plain ``.to(device)`` /
    ``torch.tensor(..., device=device)`` of pageable memory waits until everything queued on the stream has run)."""
    if t.device == torch.device(device) or torch.device(device).type == "cpu":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


class FaceCropper(nn.Module):
    """``RetinaFaceClient`` around a caller-supplied detector: ``crop_faces(images, out_size, T)`` -> (fg crops [BS, 3, *out_size],
    bg crops [N, 3, *out_size] or None, fg boxes long [BS, 4] as (x1, y1, x2, y2), confidences [BS], detected mask [BS]).
    The boxes, confidences and detected mask are what the (host-side) detector said and stay HOST tensors: the loss assembly slices
    with the boxes and branches on the other two, and every such read of a device tensor would make the host wait for the device
    (a Stage-2 micro-batch did that ~480 times, profiles/r04p_host_syncs.txt); ``to_device_async`` moves one where arithmetic needs it."""

    def __init__(self, detect_faces=no_faces):
        super().__init__()
        self.detect_faces = detect_faces

    def crop_faces(self, images_ts, out_size=(128, 128), T=20):
        H, W = images_ts.shape[2], images_ts.shape[3]
        resize = lambda c: F.interpolate(c.unsqueeze(0), size=out_size, mode="bilinear", align_corners=False)
        fg, bg, boxes, conf, found = [], [], [], [], []
        images_u8 = images_to_uint8(images_ts)
        for image, image_u8 in zip(images_ts, images_u8):
            cands = []
            for (x, y, w, h, c) in self.detect_faces(image_u8, T):
                if h <= T or w <= T:
                    continue
                x0, y0 = max(0, int(x)), max(0, int(y))
                x1, y1 = min(W, int(x + w)), min(H, int(y + h))
                if y0 + T >= y1 or x0 + T >= x1:
                    continue
                cands.append(((y1 - y0) * (x1 - x0), c, x0, y0, x1, y1))
            if not cands:
                fg.append(resize(image))
                boxes.append((0, 0, W, H))
                conf.append(0.0)
                found.append(0)
                continue
            cands.sort(key=lambda r: r[0], reverse=True)              # largest face first; the others are background faces
            crops = [resize(image[:, y0:y1, x0:x1]) for (_, _, x0, y0, x1, y1) in cands]
            fg.append(crops[0])
            bg.extend(crops[1:])
            boxes.append(cands[0][2:])
            conf.append(float(cands[0][1]))
            found.append(1)
        return (torch.cat(fg, dim=0), torch.cat(bg, dim=0) if bg else None, torch.tensor(boxes, dtype=torch.long),
                torch.tensor(conf, dtype=torch.float32), torch.tensor(found, dtype=torch.long))


GREY = (0.299, 0.587, 0.114)


def _central_box(size, ratio):
    m = int(size * (1 - ratio) / 2)
    return m, size - m


class ArcFaceWrapper(nn.Module):
    """``arcface``: the embedding network (callable on grey [N, 1, 128, 128] -> [N, 512]); ``retinaface``: a ``FaceCropper``."""

    def __init__(self, arcface, retinaface=None, dtype=torch.float16):
        super().__init__()
        self.arcface = arcface
        self.retinaface = retinaface if retinaface is not None else FaceCropper()
        self.dtype = dtype
        self._grey = {}                                             # device -> the grey-conversion weights [1, 3, 1, 1]
        for p in self.arcface.parameters():
            p.requires_grad_(False)

    def grey_weights(self, device):
        w = self._grey.get(device)
        if w is None:
            w = self._grey[device] = torch.tensor(GREY, device=device).view(1, 3, 1, 1)
        return w

    def _embed(self, grey, enable_grad):
        if getattr(self.arcface, "inference_only", False):          # an embedding module that declares it has no backward
            if enable_grad and grey.requires_grad and torch.is_grad_enabled():
                raise NotImplementedError("ArcFaceWrapper: a face was found in a tensor that requires grad, but the embedding module is "
                                          "inference_only; the alignment loss needs its input gradient")
            with torch.no_grad():
                return self.arcface(grey)
        with torch.set_grad_enabled(enable_grad):                   # any differentiable embedding module
            return self.arcface(grey)

    def embed_image_tensor(self, images_ts, T=20, embed_bg_faces=True, enable_grad=True, fg_faces_grad_mask_ratios=(1, 0.3)):
        """-> (centre-masked fg embeddings, border-masked fg embeddings, bg-face embeddings or None, boxes, confidences, detected mask);
        the first four are None when no instance has a face."""
        fg_crops, bg_crops, boxes, conf, found = self.retinaface.crop_faces(images_ts, out_size=(128, 128), T=T)
        if found.sum() == 0:
            return None, None, None, None, conf, found
        w = self.grey_weights(images_ts.device)
        grey = F.interpolate((fg_crops * w).sum(dim=1, keepdim=True).to(self.dtype), size=(128, 128), mode="bilinear", align_corners=False)
        centre_ratio, border_ratio = fg_faces_grad_mask_ratios
        grey_centre, grey_border = grey, None
        if 0 < centre_ratio < 1:                        # gradient only through the central part of the face: it must not grow
            m = torch.zeros_like(grey)
            (l, r), (t, b) = _central_box(grey.shape[3], centre_ratio), _central_box(grey.shape[2], centre_ratio)
            m[:, :, t:b, l:r] = 1
            grey_centre = gen_masked_grad_layer(m)(grey)
        if 0 < border_ratio < 1:                        # gradient only through the border: the suppression shrinks the face from outside
            m = torch.ones_like(grey)
            (l, r), (t, b) = _central_box(grey.shape[3], border_ratio), _central_box(grey.shape[2], border_ratio)
            m[:, :, t:b, l:r] = 0
            grey_border = gen_masked_grad_layer(m)(grey)
        emb_centre = self._embed(grey_centre, enable_grad)
        emb_border = self._embed(grey_border, enable_grad) if grey_border is not None else emb_centre
        emb_bg = None
        if embed_bg_faces and bg_crops is not None:
            g = F.interpolate((bg_crops * w).sum(dim=1, keepdim=True), size=(128, 128), mode="bilinear", align_corners=False).to(self.dtype)
            emb_bg = self._embed(g, enable_grad)
        return emb_centre, emb_border, emb_bg, boxes, conf, found

    def embed_reference(self, ref_images, T=20):
        """(embeddings, detected mask) of reference images: the first half of ``calc_arcface_align_loss``; a caller that aligns
        several generations to the SAME references computes it once and passes it as ``ref``."""
        ref_emb, _, _, _, _, ref_found = self.embed_image_tensor(ref_images, T, embed_bg_faces=False, enable_grad=False,
                                                                 fg_faces_grad_mask_ratios=(-1, -1))
        return ref_emb, ref_found

    def calc_arcface_align_loss(self, ref_images, aligned_images, T=20, fg_faces_grad_mask_ratios=(1, 0.3), ref=None):
        """(cosine-embedding alignment of the generated faces to the reference faces, mean squared border-masked embedding, mean squared
        background-face embedding, boxes of the generated faces or None, their confidences, their detected mask); zero losses when any
        reference instance, or every generated instance, has no face (arcface_wrapper.py:171-240).  ``ref``: ``embed_reference(ref_images)``
        when the caller already has it."""
        ref_emb, ref_found = ref if ref is not None else self.embed_reference(ref_images, T)
        emb_c, emb_b, emb_bg, boxes, conf, found = self.embed_image_tensor(aligned_images, T, embed_bg_faces=True, enable_grad=True,
                                                                           fg_faces_grad_mask_ratios=fg_faces_grad_mask_ratios)
        zero = lambda: torch.zeros((), dtype=ref_images.dtype, device=ref_images.device)
        if (1 - ref_found).sum() > 0 or found.sum() == 0:                       # host tensors: no device round trip
            return zero(), zero(), zero(), None, conf, found
        if len(ref_emb) < len(emb_c):
            ref_emb = ref_emb.repeat(len(emb_c) // len(ref_emb), 1)
        per = F.cosine_embedding_loss(ref_emb, emb_c, torch.ones(ref_emb.shape[0], device=ref_emb.device), reduction="none")
        found_d = to_device_async(found, per.device)
        n_found = found_d.sum()
        loss_align = (per * found_d).sum() / n_found
        loss_fg_suppress = ((emb_b ** 2).mean(dim=1) * found_d).sum() / n_found
        loss_bg_suppress = (emb_bg ** 2).mean() if emb_bg is not None else zero()
        return loss_align, loss_fg_suppress, loss_bg_suppress, boxes, conf, found
