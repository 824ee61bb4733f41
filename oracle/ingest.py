"""Frame ingest restated on CPU (test infrastructure -- see oracle/__init__.py).

  to_tensor_normalize  <- the two loaders of the reference:
        util_track/mp_loader.py:239-243           cv2.resize -> F.to_tensor -> F.normalize -> .to(device)   (frame stays BGR)
        perform_3D_detection_on_video_sequences.py:51-58   same with cv2.cvtColor(BGR2RGB) before to_tensor

``F.to_tensor`` / ``F.normalize`` are torchvision.transforms.functional -- a third-party dependency that is not under
/root/reference and not installed here (version unpinned in the reference).  Their published semantics, restated:
to_tensor on a uint8 HWC array = permute to CHW, convert to float32, divide by 255; normalize = subtract the
per-channel mean, divide by the per-channel std, both float32 tensors built from the Python lists.  Parity for these
two functions is therefore "unpinned" (no reference-side fixture can exist); the tests pin the device kernel to this
restatement bit for bit.  cv2.resize stays on the host in the reference and is not part of the device path.
"""
import torch

MEAN = (0.485, 0.456, 0.406)             # mp_loader.py:241, perform_3D_detection_on_video_sequences.py:57
STD = (0.229, 0.224, 0.225)


def to_tensor_normalize(frames_u8, swap_rb=False, mean=MEAN, std=STD):
    """uint8 [B,H,W,3] -> float32 [B,3,H,W].  swap_rb: the cvtColor(BGR2RGB) of the second caller."""
    x = frames_u8
    if swap_rb:
        x = x.flip(-1)
    t = x.permute(0, 3, 1, 2).contiguous().to(torch.float32).div(255)
    m = torch.as_tensor(mean, dtype=torch.float32).view(1, 3, 1, 1)
    s = torch.as_tensor(std, dtype=torch.float32).view(1, 3, 1, 1)
    return t.sub_(m).div_(s)
