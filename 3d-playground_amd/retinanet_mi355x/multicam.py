"""BASELINE configs[3] across the GPUs of a node: the tracker's per-time-step "detect every camera, then parse" loop
(MC3D_crop_tracker.py:1051-1088; camera list p1c1 .. p3c6, :1489-1509) with the cameras dealt over one process per GPU.

The reference pushes the frames of ALL cameras through one detector call on one GPU (``self.detector(self.frames,
MULTI_FRAME=True)``, :1074) and parses the result on the host.  Cameras are independent units up to the cross-camera
road-plane NMS inside ``parse_detections`` (:319-383, ``space_nms`` :618-636), which sees tens of boxes.  So the shard is the
camera, there is NO collective in the data path of the detector ("replicas only"), and the one exchange is
the hand-over of each rank's survivors -- K x (score, class, 20 box values, camera) with K <= 10 000 per call, a few hundred
on real frames -- to the rank that parses:

  rank r of N   cameras r, r + N, r + 2N, ...  (``shard``): 18 cameras over 8 GPUs = 3, 3, 2, 2, 2, 2, 2, 2
                uint8 frames -> fused ingest -> detector(MULTI_FRAME) on ITS cameras -> survivors with GLOBAL camera ids
  every rank    ``gather_detections``: one all_gather of the survivor counts, one of the padded rows (RCCL on the device
                tensors; gloo on CPU tensors in the tests)
  rank 0        ``merge``: concatenation ordered by descending score (ties: camera, then position) -- the order one
                ``batched_nms`` over all cameras returns its survivors in (D/model.py:56-57 sorts by score), independent of
                the number of ranks -- and ``parse_detections`` + ``state_to_im`` on the merged set (mc3d_post.py), exactly
                as on one GPU.

What differs from the reference's single call, by construction of any sharding: the adaptive score threshold of the
MULTI_FRAME branch (at most 10 000 candidates PER CALL, D/model.py:324-331) applies to each rank's cameras separately, so a
rank keeps at most 10 000 candidates of its 2-3 cameras where the single call keeps 10 000 of all 18.  Per-camera NMS is
unaffected (``batched_nms`` never lets boxes of different cameras suppress each other).
"""
import torch
import torch.distributed as dist

CAMERAS = ["p%dc%d" % (p, c) for p in (1, 2, 3) for c in range(1, 7)]     # MC3D_crop_tracker.py:1489-1509


def shard(n_cameras, world, rank):
    """Global camera indices of `rank`: r, r + world, ...  Every camera belongs to exactly one rank; the counts differ by at
    most one and the low ranks carry the extra camera."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside a world of %d" % (rank, world))
    return list(range(rank, n_cameras, world))


def shards(n_cameras, world):
    return [shard(n_cameras, world, r) for r in range(world)]


def to_global(local_cam_idx, my_cameras):
    """Camera index inside this rank's detector call (position in its frame batch) -> global camera index."""
    table = torch.as_tensor(my_cameras, dtype=torch.int64, device=local_cam_idx.device)
    return table[local_cam_idx.long()]


def gather_detections(scores, labels, boxes, cams, group=None):
    """Every rank's MULTI_FRAME survivors (scores [K], labels [K] int64, boxes [K,20], cams [K] GLOBAL camera ids) -> list
    over ranks of (scores, labels, boxes, cams), on every rank, on the inputs' device.  Two collectives: the counts, then the
    rows padded to the largest count (floats and integers packed side by side: scores | boxes as fp32 bits, labels, cams)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return [(scores, labels, boxes, cams)]
    dev = scores.device
    out_dev = dev
    if dev.type == "cuda" and dist.get_backend(group) == "gloo":
        dev = torch.device("cpu")        # one-GPU rehearsal (RN_REHEARSE_ONE_GPU): gloo gathers host tensors only
    k = torch.tensor([scores.numel()], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(k) for _ in range(world)]
    dist.all_gather(counts, k, group=group)
    counts = [int(c) for c in counts]                                    # one host read: the parse on rank 0 needs the sizes anyway
    kmax = max(counts)
    if kmax == 0:
        empty = (scores[:0], labels[:0], boxes[:0].reshape(0, 20), cams[:0])
        return [empty for _ in range(world)]
    row = torch.zeros((kmax, 23), dtype=torch.int32, device=dev)
    n = scores.numel()
    if n:
        row[:n, 0] = scores.float().contiguous().view(torch.int32).to(dev)
        row[:n, 1:21] = boxes.float().contiguous().view(torch.int32).view(n, 20).to(dev)
        row[:n, 21] = labels.to(torch.int32).to(dev)
        row[:n, 22] = cams.to(torch.int32).to(dev)
    rows = [torch.empty_like(row) for _ in range(world)]
    dist.all_gather(rows, row, group=group)
    out = []
    for r, c in zip(rows, counts):
        r = r[:c].to(out_dev)
        out.append((r[:, 0].contiguous().view(torch.float32), r[:, 21].long(),
                    r[:, 1:21].contiguous().view(torch.float32).view(c, 20), r[:, 22].long()))
    return out


def merge(parts):
    """Concatenate the ranks' survivors and order them as ONE ``batched_nms`` over all cameras would have returned them: by
    descending score (D/model.py:56-57), ties by camera and then by the position inside that camera's call -- two stable
    sorts, camera first.  The result does not depend on how many ranks the cameras were dealt over."""
    scores = torch.cat([p[0] for p in parts])
    labels = torch.cat([p[1] for p in parts])
    boxes = torch.cat([p[2].reshape(-1, 20) for p in parts])
    cams = torch.cat([p[3] for p in parts])
    by_cam = torch.sort(cams, stable=True)[1]
    order = by_cam[torch.sort(scores[by_cam], descending=True, stable=True)[1]]
    return scores[order], labels[order], boxes[order], cams[order]
