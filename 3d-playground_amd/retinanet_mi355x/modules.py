"""nn.Module surface of the detector, mirroring the reference's classes name for name.

  ResNet / resnet18..152      <- D/model.py:208-453 (directional, 12 regression outputs, 3 losses, 3 eval modes)
                                  R/model.py:167-367 (2D twin: 4 regression outputs, 2 losses, ClipBoxes, 0.05 threshold)
  PyramidFeatures, RegressionModel, ClassificationModel, BasicBlock, Bottleneck
                               <- parameter containers with the reference's attribute names, so ``state_dict()``
                                  keys / shapes are identical (R18 156, R50 354, R101 660 entries) and callers can
                                  keep doing ``net.classificationModel.output.weight = nn.Parameter(...)``
                                  (train_detector_3D_angle.py:290-291)
  Anchors, BBoxTransform, ClipBoxes, FocalLoss, calc_iou   <- standalone modules of D/anchors.py, D/utils.py,
                                  D/losses.py (and their R/ twins)

All arithmetic runs in ``engine.Engine`` / ``ops`` on the MI355X; parameters stay plain OIHW fp32
``nn.Parameter``s (packed copies are a derived cache).  Batch-norm is always evaluated with its running
statistics: the reference freezes it in ``__init__`` and after every ``.train()`` (D/model.py:260, 278-282;
train_detector_3D_angle.py:332-334); gamma / beta still receive gradients.
"""
import math

import torch
import torch.nn as nn

from . import _hip, arch, engine, ops, torch_ops


# ----------------------------------------------------------------------------------------------- containers
class BasicBlock(nn.Module):                      # D/utils.py:12-43
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


class Bottleneck(nn.Module):                      # D/utils.py:46-80
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


class PyramidFeatures(nn.Module):                 # D/model.py:59-82
    def __init__(self, C3_size, C4_size, C5_size, feature_size=256):
        super().__init__()
        self.P5_1 = nn.Conv2d(C5_size, feature_size, 1, 1, 0)
        self.P5_upsampled = nn.Upsample(scale_factor=2, mode="nearest")
        self.P5_2 = nn.Conv2d(feature_size, feature_size, 3, 1, 1)
        self.P4_1 = nn.Conv2d(C4_size, feature_size, 1, 1, 0)
        self.P4_upsampled = nn.Upsample(scale_factor=2, mode="nearest")
        self.P4_2 = nn.Conv2d(feature_size, feature_size, 3, 1, 1)
        self.P3_1 = nn.Conv2d(C3_size, feature_size, 1, 1, 0)
        self.P3_2 = nn.Conv2d(feature_size, feature_size, 3, 1, 1)
        self.P6 = nn.Conv2d(C5_size, feature_size, 3, 2, 1)
        self.P7_1 = nn.ReLU()
        self.P7_2 = nn.Conv2d(feature_size, feature_size, 3, 2, 1)


class _Tower(nn.Module):
    def __init__(self, num_features_in, n_out, feature_size):
        super().__init__()
        for i in range(1, 5):
            setattr(self, "conv%d" % i, nn.Conv2d(num_features_in if i == 1 else feature_size, feature_size, 3, padding=1))
            setattr(self, "act%d" % i, nn.ReLU())
        self.output = nn.Conv2d(feature_size, n_out, 3, padding=1)


class RegressionModel(_Tower):                    # D/model.py:120-137
    def __init__(self, num_features_in, num_anchors=9, feature_size=256, n_outputs=8):
        super().__init__(num_features_in, num_anchors * n_outputs, feature_size)
        self.n_outputs = n_outputs


class ClassificationModel(_Tower):                # D/model.py:160-180
    def __init__(self, num_features_in, num_anchors=9, num_classes=80, prior=0.01, feature_size=256):
        super().__init__(num_features_in, num_anchors * num_classes, feature_size)
        self.num_classes = num_classes
        self.num_anchors = num_anchors
        self.output_act = nn.Sigmoid()


# ----------------------------------------------------------------------------------------------- standalone modules
class Anchors(nn.Module):                         # D/anchors.py:6-40
    def __init__(self, pyramid_levels=None, strides=None, sizes=None, ratios=None, scales=None):
        super().__init__()
        if not (pyramid_levels is None and strides is None and sizes is None and ratios is None and scales is None):
            raise NotImplementedError("the reference ignores every non-default argument except by crashing "
                                      "(D/anchors.py:9-19 only assigns the attributes in the None branches)")
        self.pyramid_levels = [3, 4, 5, 6, 7]
        self.strides = [2 ** x for x in self.pyramid_levels]
        self.sizes = [2 ** (x + 2) for x in self.pyramid_levels]

    def forward(self, image):
        if not image.is_cuda:
            raise RuntimeError("anchors are generated on the MI355X (no CPU fallback); got device %s" % image.device)
        return torch.ops.retinanet_mi355x.anchors(image.shape[2], image.shape[3], image.device)


class BBoxTransform(nn.Module):                   # D/utils.py:82-149 / R/utils.py:82-126
    def __init__(self, mean=None, std=None, directional=True):
        super().__init__()
        if mean is not None or std is not None:
            raise NotImplementedError("custom mean/std: the directional transform never reads them (D/utils.py:102-149) "
                                      "and every caller of the 2D one uses the defaults (R/model.py:196)")
        self.directional = directional

    def forward(self, boxes, regression):
        if self.directional:
            return torch.ops.retinanet_mi355x.decode_dir(boxes, regression)
        return torch.ops.retinanet_mi355x.decode_2d(boxes, regression, False, 0, 0)


class ClipBoxes(nn.Module):                       # R/utils.py:129-144
    def __init__(self, width=None, height=None):
        super().__init__()

    def forward(self, boxes, img):
        ops.clip_boxes_check(boxes)
        torch.ops.retinanet_mi355x.clip_boxes_(boxes, img.shape[2], img.shape[3])
        return boxes


class FocalLoss(nn.Module):                       # D/losses.py:24-362 / R/losses.py:24-177
    def __init__(self, directional=True):
        super().__init__()
        self.directional = directional

    def forward(self, classifications, regressions, anchors, annotations):
        return torch_ops.focal_loss(classifications, regressions, anchors, annotations, self.directional)


def calc_iou(a, b):                               # D/losses.py:5-22
    return torch.ops.retinanet_mi355x.pairwise_iou(a, b)


# ----------------------------------------------------------------------------------------------- the network
class _NetFn(torch.autograd.Function):
    """The whole training forward (backbone + FPN + heads + anchors + loss) as one autograd node."""

    @staticmethod
    def forward(ctx, net, img, ann, *params):
        eng = net._engine
        P = net._tensor_dict()
        reg, cls, S = eng.forward(P, img, save=True)
        anc = eng.anchors(img.shape[2], img.shape[3], img.device)
        lib = _hip.load()
        ann_c = _hip.f32c(ann)
        B, A, C = cls.shape
        ws = ops.focal_workspace(B, A, img.device)
        losses = torch.empty(3, dtype=torch.float32, device=img.device)
        _hip.check(lib.rn_focal_loss_fwd(cls.data_ptr(), reg.data_ptr(), anc.data_ptr(), _hip.ptr(ann_c), B, A, C,
                                         ann_c.shape[1], int(net.directional), ws.data_ptr(), losses.data_ptr(),
                                         _hip.stream()), "rn_focal_loss_fwd")
        ctx.net, ctx.S, ctx.pack = net, S, (reg, cls, anc, ann_c, ws)
        return losses[0:1], losses[1:2], losses[2:3]

    @staticmethod
    def backward(ctx, g_cls, g_reg, g_vp):
        net, S = ctx.net, ctx.S
        reg, cls, anc, ann_c, ws = ctx.pack
        ctx.S = ctx.pack = None
        lib = _hip.load()
        B, A, C = cls.shape
        g = torch.cat([t.reshape(1).float() for t in (g_cls, g_reg, g_vp)])
        dcls, dreg = torch.empty_like(cls), torch.empty_like(reg)
        _hip.check(lib.rn_focal_loss_bwd(cls.data_ptr(), reg.data_ptr(), anc.data_ptr(), _hip.ptr(ann_c), B, A, C,
                                         ann_c.shape[1], int(net.directional), ws.data_ptr(), g.data_ptr(),
                                         dcls.data_ptr(), dreg.data_ptr(), _hip.stream()), "rn_focal_loss_bwd")
        red = net.__dict__.get("_reducer")
        eng = net._engine
        if red is not None:
            red.backward_begins()
        grads = eng.backward(S, dreg, dcls, cls)
        if red is not None:
            red.finalize_flat(eng._flat["arena"])          # waits for the buckets; grads are views of that buffer
        return (None, None, None) + tuple(grads[n] for n in eng.param_names)


class ResNet(nn.Module):
    """Drop-in for the reference's ``ResNet`` (D/model.py:208 / R/model.py:167)."""

    def __init__(self, num_classes, block, layers, directional=True):
        self.inplanes = 64
        super().__init__()
        self.directional = directional
        self.ingest_swap_rb = False                 # uint8 frame input: BGR->RGB swap of perform_3D_detection...py:52
        self.ingest_mean, self.ingest_std = ops.IMAGENET_MEAN, ops.IMAGENET_STD
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        e = block.expansion
        self.fpn = PyramidFeatures(128 * e, 256 * e, 512 * e)
        n_reg = 12 if directional else 4
        self.regressionModel = RegressionModel(256, n_outputs=n_reg)
        self.classificationModel = ClassificationModel(256, num_classes=num_classes)
        self.anchors = Anchors()
        self.regressBoxes = BBoxTransform(directional=directional)
        self.clipBoxes = ClipBoxes()
        self.focalLoss = FocalLoss(directional=directional)
        for m in self.modules():                                           # D/model.py:244-250
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        prior = 0.01                                                       # D/model.py:252-258
        self.classificationModel.output.weight.data.fill_(0)
        self.classificationModel.output.bias.data.fill_(-math.log((1.0 - prior) / prior))
        self.regressionModel.output.weight.data.fill_(0)
        self.regressionModel.output.bias.data.fill_(0)
        self.freeze_bn()
        arch_name = {(BasicBlock, (2, 2, 2, 2)): "resnet18", (BasicBlock, (3, 4, 6, 3)): "resnet34",
                     (Bottleneck, (3, 4, 6, 3)): "resnet50", (Bottleneck, (3, 4, 23, 3)): "resnet101",
                     (Bottleneck, (3, 8, 36, 3)): "resnet152"}.get((block, tuple(layers)))
        if arch_name is None:
            raise ValueError("Block type %s / layers %s not understood" % (block, layers))
        self.__dict__["_arch_name"] = arch_name
        self.__dict__["_engine"] = engine.Engine(arch_name, num_classes, n_reg)

    def _make_layer(self, block, planes, blocks, stride=1):                # D/model.py:262-276
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def freeze_bn(self):
        """Freeze BatchNorm layers (D/model.py:278-282).  The engine always uses running statistics."""
        for layer in self.modules():
            if isinstance(layer, nn.BatchNorm2d):
                layer.eval()

    def set_gradient_reducer(self, reducer):
        """Attach a ``ddp.GradReducer``: gradients are then averaged over the process group inside backward,
        bucket by bucket as the reverse schedule finishes layers (replaces nn.DataParallel,
        train_detector_3D_angle.py:317).  None detaches it."""
        self.__dict__["_reducer"] = reducer
        if reducer is not None:
            reducer.attach(self._engine)
        else:
            self._engine.bucket_hook = None
            self._engine.set_flat_grads(None)

    def set_compute_dtype(self, dtype):
        """"fp32" (default: the reference's arithmetic), "fp8" (inference only, after calibrate_fp8) or "bf16" (BASELINE configs[2]): bf16 activations and packed weights
        on the bf16 matrix cores, fp32 accumulation, parameters, gradients and loss.  Not the reference's numerics: expect
        ~1e-2 relative differences in losses and gradients (tests/test_gpu_model_bf16.py)."""
        old = self._engine
        eng = engine.Engine(self._arch_name, old.num_classes, old.n_reg, dtype=dtype)
        eng.set_flat_grads(old.flat_bucket_bytes, old.flat_tail_bytes)    # the attached reducer's bucket layout, tail included
        eng.bucket_hook = old.bucket_hook
        eng.fp8_scales = getattr(old, "fp8_scales", None)
        self.__dict__["_engine"] = eng
        return self

    def calibrate_fp8(self, frames, margin=1.0):
        """BASELINE configs[4]: run the fp32 engine (direct kernels; whatever engine the model is on now, bf16 included, the
        calibration pass switches to it) on representative frames [B,3,H,W], record the magnitude of every activation tensor,
        and switch the model to the fp8 inference engine with those per-tensor scales (e4m3 activations and
        per-output-channel-scaled e4m3 weights on the fp8 MFMA; csrc/conv_fp8.hip).  Inference only: ``net.train()`` forwards
        raise.  ``set_compute_dtype("fp32")`` switches back (the scales are kept)."""
        if self._engine.fp8 or self._engine.bf16:
            self.set_compute_dtype("fp32")
        scales = self._engine.calibrate(self._tensor_dict(), frames, margin)
        self.set_compute_dtype("fp8")
        self._engine.fp8_scales = scales
        return scales

    def use_flat_gradients(self, on=True):
        """Write parameter gradients into one persistent device buffer (stable ``p.grad`` pointers from step to step,
        no per-step allocations; see Engine.set_flat_grads).  Implied by set_gradient_reducer.  Leave it off if your
        loop keeps references to ``p.grad`` tensors across ``zero_grad(set_to_none=True)``: the next backward
        overwrites that memory."""
        self._engine.set_flat_grads((32 << 20) if on else None)

    def _replicate_for_data_parallel(self):
        # nn.DataParallel (train_detector_3D_angle.py:317) replicates the module onto device threads; the replicas
        # would share ONE engine (packed-weight caches, gradient arena, Winograd workspaces) across devices.
        raise RuntimeError(
            "this drop-in does not run under torch.nn.DataParallel: its HIP engine keeps per-device state and would be "
            "shared by the replica threads.  Use one process per GPU instead: retinanet_mi355x.ddp.init_from_env() + "
            "net.set_gradient_reducer(ddp.GradReducer()) (see INTEGRATION.md, 'Multi-GPU').")

    def load_state_dict(self, state_dict, strict=True, **kwargs):
        """nn.Module.load_state_dict, also for the checkpoints the reference's multi-GPU trainer writes: it saves
        ``nn.DataParallel(model).state_dict()`` (train_detector_3D_angle.py:416-417), whose every key carries a ``module.``
        prefix, and strips that prefix itself before loading (``to_cpu``, train_detector_3D_angle.py:39-59).  A dict in which
        EVERY key has the prefix is taken as such a checkpoint and loaded without it; anything else goes through unchanged, so
        ``strict`` keeps its meaning (a half-prefixed dict still reports its unexpected / missing keys)."""
        keys = list(state_dict.keys())
        if keys and all(k.startswith("module.") for k in keys):
            stripped = type(state_dict)()                  # keeps an OrderedDict an OrderedDict
            for k in keys:
                stripped[k[len("module."):]] = state_dict[k]
            meta = getattr(state_dict, "_metadata", None)
            if meta is not None:                           # per-module version records, keyed "module.<path>" / "module"
                stripped._metadata = type(meta)((k[len("module."):] if k.startswith("module.") else ("" if k == "module" else k), v)
                                                for k, v in meta.items())
            state_dict = stripped
        return super().load_state_dict(state_dict, strict=strict, **kwargs)

    def _tensor_dict(self):
        d = dict(self.named_parameters())
        d.update(dict(self.named_buffers()))
        return d

    def forward(self, inputs, LOCALIZE=False, MULTI_FRAME=False):
        eng = self._engine
        if self.training:
            img_batch, annotations = inputs
            ops.check_labels(annotations, self.directional)
            P = dict(self.named_parameters())
            params = [P[n] for n in eng.param_names]
            out = _NetFn.apply(self, img_batch, annotations, *params)
            return out if self.directional else out[:2]
        img_batch = inputs
        if MULTI_FRAME and not self.directional:
            raise TypeError("forward() got an unexpected keyword argument 'MULTI_FRAME'")   # R/model.py:243
        with torch.no_grad():
            if img_batch.dtype == torch.uint8:
                # Extension (the reference's model cannot take it): raw uint8 [B,H,W,3] frames.  to_tensor + normalize
                # of the loaders (util_track/mp_loader.py:239-243) run fused on device, straight into the stem's layout.
                x4 = ops.frame_ingest(img_batch, swap_rb=self.ingest_swap_rb, mean=self.ingest_mean, std=self.ingest_std,
                                      nhwc4=True)
                H_in, W_in = x4.shape[1], x4.shape[2]
                reg, cls, _ = eng.forward(self._tensor_dict(), None, save=False, x4=x4)
            else:
                H_in, W_in = img_batch.shape[2], img_batch.shape[3]
                reg, cls, _ = eng.forward(self._tensor_dict(), img_batch, save=False)
            anc = eng.anchors(H_in, W_in, reg.device)
            if self.directional:
                if MULTI_FRAME:                                            # D/model.py:311-344, decoding the survivors only
                    return ops.detect_multi(cls, reg, anc)
                if LOCALIZE:                                               # D/model.py:362-363
                    return ops.decode_dir(anc, reg), cls
                return ops.detect_single(cls, reg, anc)                    # D/model.py:365-397
            boxes = ops.decode_2d(anc, reg, clip_hw=(H_in, W_in))          # R/model.py:270-271
            if LOCALIZE:
                return boxes, cls
            return ops.postprocess_2d(cls, boxes)                          # R/model.py:285-311


def _load_pretrained(model, name):
    raise RuntimeError("pretrained=True fetches %s over the network in the reference (D/model.py:10-16, 408); there is "
                       "no network here -- build with pretrained=False and load_state_dict() a local checkpoint "
                       "(strict=False for an ImageNet backbone)" % name)


def _make(name, block, layers):
    def ctor(num_classes, pretrained=False, directional=True, **kwargs):
        model = ResNet(num_classes, block, layers, directional=directional, **kwargs)
        if pretrained:
            _load_pretrained(model, name)
        return model
    ctor.__name__ = name
    ctor.__doc__ = "Constructs a %s model (D/model.py:401-453)." % name
    return ctor


resnet18 = _make("resnet18", BasicBlock, [2, 2, 2, 2])
resnet34 = _make("resnet34", BasicBlock, [3, 4, 6, 3])
resnet50 = _make("resnet50", Bottleneck, [3, 4, 6, 3])
resnet101 = _make("resnet101", Bottleneck, [3, 4, 23, 3])
resnet152 = _make("resnet152", Bottleneck, [3, 8, 36, 3])
