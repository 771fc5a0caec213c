"""Host-side pieces of the reference's test driver that sit right after the forward pass:
the crop of the padding added by generate_test_cbmv, the PFM writer / reader, the EPE / bad-x metric and the checkpoint
key fix-up.  Host NumPy as in the reference, plus the metric on the device (msnet_epe_badx) when it is handed GPU tensors;
file decoding (cv2.imread) and dataset lists stay with the caller.

  crop_disparity      main_msnet.py:585-589
  save_pfm / read_pfm src/utils/pfmutil.py:86-110 / :48-83   (same bytes: header 'Pf', 'W H', signed scale, rows bottom-up)
  get_epe_rate        main_msnet.py:708-713      (NumPy inputs: host; GPU tensors: csrc/metrics.hip)
  strip_module_prefix keys saved from nn.DataParallel carry 'module.' (main_msnet.py:174, 509-526)
"""
import re
import sys

import numpy as np


def crop_disparity(disp, crop_height, crop_width, height, width):
    """disp [N, crop_height, crop_width] (or torch tensor) -> the [height, width] map of sample 0: the padding was added on
    the TOP and the RIGHT (cbmv_generator.py:780-788), so rows crop_height-height.. and columns 0..width are kept."""
    if hasattr(disp, "detach"):
        disp = disp.detach().cpu().numpy()
    if height <= crop_height and width <= crop_width:
        return disp[0, crop_height - height: crop_height, 0:width]
    return disp[0, :, :]


def save_pfm(fname, image, scale=1):
    """pfmutil.save: float32 [H,W], [H,W,1] (header 'Pf') or [H,W,3] ('PF'); rows are written bottom-up; the scale's sign
    encodes the byte order (negative = little endian)."""
    image = np.asarray(image)
    if image.dtype.name != "float32":
        raise Exception("save_pfm writes float32 data only (got dtype %s)" % image.dtype.name)
    if image.ndim == 3 and image.shape[2] == 3:
        color = True
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        color = False
    else:
        raise Exception("save_pfm takes an [H,W], [H,W,1] or [H,W,3] array (got shape %s)" % (image.shape,))
    endian = image.dtype.byteorder
    if endian == "<" or (endian == "=" and sys.byteorder == "little"):
        scale = -scale
    with open(fname, "wb") as f:
        f.write(b"PF\n" if color else b"Pf\n")
        f.write(("%d %d\n" % (image.shape[1], image.shape[0])).encode("latin-1"))
        f.write(("%f\n" % scale).encode("latin-1"))
        np.flipud(image).tofile(f)


def read_pfm(fname):
    """pfmutil.readPFM: -> float32 [H, W] (single channel) or [H, W, 3], rows top-down again."""
    with open(fname, "rb") as f:
        kind = f.readline().decode("latin-1")
        if "PF" in kind:
            channels = 3
        elif "Pf" in kind:
            channels = 1
        else:
            raise ValueError("not a PFM file: %r" % kind)
        width, height = (int(v) for v in re.findall(r"\d+", f.readline().decode("latin-1")))
        big_endian = "-" not in f.readline().decode("latin-1")
        data = np.frombuffer(f.read(width * height * channels * 4), dtype=(">f4" if big_endian else "<f4"))
    shape = (height, width) if channels == 1 else (height, width, 3)
    return np.flipud(data.reshape(shape)).astype(np.float32)


def get_epe_rate(disp, prediction, max_disp=192, threshold=3.0):
    """End-point error and bad-`threshold` rate over the pixels with 0.001 <= gt <= max_disp (main_msnet.py:708-713).
    NumPy arrays / CPU tensors are evaluated on the host as in the reference; when either operand is a GPU tensor the other is
    uploaded and both are reduced on the device (msnet_epe_badx: one HBM pass, 24 bytes back), so a GPU disparity map never
    has to be copied to the host for the metric."""
    d_gpu, p_gpu = bool(getattr(disp, "is_cuda", False)), bool(getattr(prediction, "is_cuda", False))
    if d_gpu or p_gpu:
        # the usual driver case is a GPU prediction with a ground truth read from a PFM file (NumPy / CPU tensor): the host
        # side is uploaded to the prediction's device; two host-side operands never touch the GPU
        import torch
        dev = (prediction if p_gpu else disp).device
        up = lambda a: a if bool(getattr(a, "is_cuda", False)) else torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(dev)   # noqa: E731
        return get_epe_rate_gpu(up(disp), up(prediction), max_disp, threshold)
    if hasattr(disp, "detach"):
        disp = disp.detach().numpy()
    if hasattr(prediction, "detach"):
        prediction = prediction.detach().numpy()
    mask = np.logical_and(disp >= 0.001, disp <= max_disp)
    err = np.abs(prediction[mask] - disp[mask])
    return np.mean(err), np.sum(err > threshold) / np.sum(mask)


def get_epe_rate_gpu(disp, prediction, max_disp=192, threshold=3.0):
    import torch
    from . import _lib
    gt = _lib.require_gpu_f32(disp, "disp")
    pred = _lib.require_gpu_f32(prediction, "prediction")
    if gt.shape != pred.shape:
        raise ValueError("ground truth %s and prediction %s differ in shape" % (tuple(gt.shape), tuple(pred.shape)))
    out = torch.empty(3, dtype=torch.float64, device=gt.device)
    _lib.check(_lib.load().msnet_epe_badx(_lib.ptr(gt), _lib.ptr(pred), gt.numel(), float(max_disp), float(threshold), _lib.ptr(out),
                                          _lib.stream_ptr()), "msnet_epe_badx")
    s, bad, valid = (float(v) for v in out.cpu())
    if valid == 0:
        return float("nan"), float("nan")              # np.mean of an empty selection, as the reference would produce
    return s / valid, bad / valid


def strip_module_prefix(state_dict):
    """Checkpoints written from nn.DataParallel: 'module.conv3dbn_1.0.weight' -> 'conv3dbn_1.0.weight'."""
    return {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
