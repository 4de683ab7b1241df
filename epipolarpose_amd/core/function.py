"""Training loop -- mirror of the reference's ``lib/core/function.py:14-63`` (``train_integral``).

Same signature and logging ("Speed ... samples/s").  Differences that matter on MI355X:
* the self-supervision step the reference defines but never wires in is inserted between the
  forward pass and the criterion when ``config.DATASET.TRI`` is set: labels come from
  ``self_supervision_device(preds.detach(), meta)`` and never leave the GPU;
* the loss is accumulated on the device and read back only every ``PRINT_FREQ`` iterations (the reference
  synchronises with ``loss.item()`` every step, function.py:48);
* the backbone runs under bf16 autocast when ``train_integral.autocast`` is true (default).
"""
import logging
import time

import numpy as np
import torch

from .. import hip
from ..dataset.collate import tri_batch_to_view_major
from ..utils.img_utils import self_supervision_device
from ..utils.utils import AverageMeter
from .integral_loss import joint_location_result_device

logger = logging.getLogger(__name__)


def train_step(model, criterion, optimizer, batch_data, batch_label, batch_label_weight, meta=None, n_view=None,
               ss_method="iterative", autocast=True, grad_sync=None):
    """One optimisation step on device-resident tensors.  Returns the (device) loss tensor."""
    if grad_sync is not None:
        grad_sync.zero_grad()                 # gradients live in the flat all-reduce buckets
    else:
        optimizer.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        preds = model(batch_data)
    if meta is not None and n_view:          # self-supervised: pseudo labels by multi-view triangulation
        batch_label, batch_label_weight = self_supervision_device(preds.detach(), meta, n_view=n_view, method=ss_method,
                                                                  num_joints=getattr(criterion, "num_joints", None))
    loss = criterion(preds, batch_label, batch_label_weight)
    loss.backward()
    if grad_sync is not None:
        grad_sync.finish()
    optimizer.step()
    return loss.detach()


class GraphedTrainStep:
    """One optimisation step captured as a hipGraph and replayed (north-star: HIP graphs instead of a tracing compiler).

    Forward, criterion, backward and Adam are captured once on static tensors and replayed with a single host call; inputs
    are copied into the static buffers before each replay.  Requires a capturable optimizer
    (``torch.optim.Adam(..., capturable=True)``) and warmed-up MIOpen kernels (done here on a side stream).
    Measured on MI355X / ROCm 7.2 ("measured and rejected"): replay adds ~2.5 us per graph node, 11.4 ms per
    step against 9.5 ms for eager launches at ~570 launches per step; the eager step is GPU-bound (host enqueue 3-4 ms
    against 7.2 ms, kernels back to back with 0.00 us gaps in the trace) and overlaps its weight gradients on a second
    stream -- so ``bench.py`` keeps eager as the default and this class as an option (``--graph 1``).
    """

    def __init__(self, model, criterion, optimizer, batch_data, batch_label, batch_label_weight, meta=None, n_view=None,
                 ss_method="iterative", autocast=True, warmup=3):
        self.data, self.label, self.weight = batch_data, batch_label, batch_label_weight
        # a captured step is one stream of nodes: the eager path's second stream and its end-of-pass slab sum (host-side table uploads
        # with an event wait) stay out of the capture
        self._modes = (hip.glue().wgrad_stream_mode(0), hip.glue().defer_wgrad_reduce(False))
        args = (model, criterion, optimizer, self.data, self.label, self.weight)
        kw = dict(meta=meta, n_view=n_view, ss_method=ss_method, autocast=autocast)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                train_step(*args, **kw)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        try:
            with torch.cuda.graph(self.graph):
                self.loss = train_step(*args, **kw)
        finally:
            # the process-wide modes only have to hold while the launches are being RECORDED: replays repeat the captured launches
            # whatever the modes are, and an eager train_step afterwards (bench A/B of graph vs eager) gets its second stream back
            hip.glue().wgrad_stream_mode(self._modes[0])
            hip.glue().defer_wgrad_reduce(self._modes[1])

    def __call__(self, batch_data=None, batch_label=None, batch_label_weight=None):
        if batch_data is not None and batch_data is not self.data:
            self.data.copy_(batch_data, non_blocking=True)
        if batch_label is not None and batch_label is not self.label:
            self.label.copy_(batch_label, non_blocking=True)
        if batch_label_weight is not None and batch_label_weight is not self.weight:
            self.weight.copy_(batch_label_weight, non_blocking=True)
        self.graph.replay()
        return self.loss


def train_integral(config, train_loader, model, criterion, optimizer, epoch, grad_sync=None):
    batch_time, data_time, losses = AverageMeter(), AverageMeter(), AverageMeter()
    model.train()
    use_ss = bool(config.DATASET.TRI)
    n_view = 2 if use_ss else None           # reference pairing: first / second half of the batch (img_utils.py:194)
    pending, pending_n = None, 0
    end = time.time()
    for i, data in enumerate(train_loader):
        data_time.update(time.time() - end)
        if isinstance(data, dict):            # stock default_collate of TRI items: {'cam_1': bundle, 'cam_2': bundle} -> view-major
            data = tri_batch_to_view_major(data)
        batch_data, batch_label, batch_label_weight, meta = data
        batch_data = batch_data.cuda(non_blocking=True)
        batch_label = batch_label.cuda(non_blocking=True)
        batch_label_weight = batch_label_weight.cuda(non_blocking=True)
        batch_size = batch_data.size(0)
        loss = train_step(model, criterion, optimizer, batch_data, batch_label, batch_label_weight,
                          meta=hip.DeviceMeta(meta, batch_data.device) if use_ss else None, n_view=n_view, grad_sync=grad_sync)
        pending = loss * batch_size if pending is None else pending + loss * batch_size
        pending_n += batch_size
        if i % config.PRINT_FREQ == 0:
            val = loss.item()                 # the only host synchronisation
            losses.update(pending.item() / pending_n, pending_n)
            losses.val = val
            pending, pending_n = None, 0
            batch_time.update(time.time() - end)
            msg = 'Epoch: [{0}][{1}/{2}]\t' \
                  'Time {batch_time.val:.3f}s ({batch_time.avg:.3f}s)\t' \
                  'Speed {speed:.1f} samples/s\t' \
                  'Data {data_time.val:.3f}s ({data_time.avg:.3f}s)\t' \
                  'Loss {loss.val:.5f} ({loss.avg:.5f})'.format(
                      epoch, i, len(train_loader), batch_time=batch_time, speed=batch_size / max(batch_time.val, 1e-9),
                      data_time=data_time, loss=losses)
            logger.info(msg)
        else:
            batch_time.update(time.time() - end)
        end = time.time()
    return losses.avg


def validate_integral(val_loader, model, num_joints=None):
    """function.py:66-110: run the network over the validation loader and decode every sample.

    Same contract as the reference -- returns ``preds_in_patch_with_score``: float64 ndarray [len(dataset), J, 4] (x, y, z
    in 256-pixel patch units, score 1) -- but the soft-argmax decode stays on the GPU and the host sees ONE copy at the
    end (the reference copies and reshapes per batch and patches ragged last batches by hand, :92-106)."""
    print("Validation stage")
    model.eval()
    chunks = []
    with torch.no_grad():
        for data in val_loader:
            batch_data = data[0].cuda(non_blocking=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                preds = model(batch_data)
            chunks.append(joint_location_result_device(256, 256, preds, num_joints))        # [B, 3J] f32, normalised
    xyz = torch.cat(chunks, dim=0)[:len(val_loader.dataset)]
    coords = xyz.cpu().numpy().astype(float)
    coords = coords.reshape((coords.shape[0], coords.shape[1] // 3, 3))
    coords[:, :, 0] = (coords[:, :, 0] + 0.5) * 256                                          # integral_loss.py:199-201
    coords[:, :, 1] = (coords[:, :, 1] + 0.5) * 256
    coords[:, :, 2] = coords[:, :, 2] * 256
    return np.concatenate((coords, np.ones((coords.shape[0], coords.shape[1], 1))), axis=2)


def eval_integral(epoch, preds_in_patch_with_score, val_loader, final_output_path, debug=False):
    """function.py:113-135: patch -> original-image coordinates for every sample (scale 1, rotation 0, 2000 mm box), then
    ``dataset.evaluate``.  The per-sample Python loop over ``trans_coords_from_patch_to_org_3d`` becomes one
    ``epi_decode_to_image`` launch over the whole validation set."""
    print("Evaluation stage")
    imdb = val_loader.dataset
    db = imdb.db
    n = len(imdb)
    p = np.asarray(preds_in_patch_with_score, dtype=np.float64)[:n]
    dev = torch.device("cuda", torch.cuda.current_device())
    xyz = np.stack([p[:, :, 0] / 256 - 0.5, p[:, :, 1] / 256 - 0.5, p[:, :, 2] / 256], axis=2).reshape(n, -1)
    meta = {"center_x": np.array([r["center_x"] for r in db], dtype=np.float64), "center_y": np.array([r["center_y"] for r in db], dtype=np.float64),
            "width": np.array([r["width"] for r in db], dtype=np.float64), "height": np.array([r["height"] for r in db], dtype=np.float64),
            "scale": np.ones(n), "rot": np.zeros(n)}
    xyz32 = xyz.astype(np.float32)
    if np.array_equal(xyz32.astype(np.float64), xyz):
        # predictions that come from validate_integral are float32 values (the soft-argmax output): the device decode is lossless
        kps = hip.decode_to_image(torch.from_numpy(xyz32).to(dev), hip.DeviceMeta(meta, dev), 256.0, 256.0, 2000.0).cpu().numpy()
    else:
        # genuinely float64 predictions: the reference decodes them in float64 (function.py:122-128) -- so do we, on the host
        from ..utils.img_utils import trans_coords_from_patch_to_org_3d
        kps = np.stack([trans_coords_from_patch_to_org_3d(p[i, :, :3].copy(), meta["center_x"][i], meta["center_y"][i], meta["width"][i],
                                                          meta["height"][i], 256, 256, 2000., 2000.) for i in range(n)])
    preds_in_img = np.concatenate([kps, p[:, :, 3:4]], axis=2)
    name_value, perf = imdb.evaluate(preds_in_img.copy(), final_output_path, debug=debug)
    for name, value in name_value:
        logger.info('Epoch[%d] Validation-%s %f', epoch, name, value)
    return perf
