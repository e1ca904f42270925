"""Evaluation metrics written by the CLI (run/test_sdm.py:185-208), without pytorch3d / open3d.

chamfer   : pytorch3d.loss.chamfer_distance defaults -- mean squared nearest-neighbour distance, both directions
            summed, averaged over the batch (reference call site run/test_sdm.py:186, evaluated on the device).
f-score   : util/evaluation.py:28-52 (threshold 0.1 on nearest-neighbour Euclidean distances, open3d
            compute_point_cloud_distance).
emd       : util/evaluation.py:5-11 (Hungarian assignment on the Euclidean distance matrix; host, scipy -- as the reference).
accuracy  : util/evaluation.py:13-26 (precision@k in percent).

The nearest-neighbour searches under chamfer / F-score run in liblsdm_hip.so (``lsdm_nn_distance``, csrc/metrics.hip)
whenever the point sets are device tensors -- the CLI's HIP backend keeps the sample on the device, as the reference
does (``pred = sample.float().to(device)``).  Host tensors (the reference moves them to the CPU for emd / F-score)
are evaluated from the definitions with torch; device tensors never fall back to that.
"""
import torch
from scipy.optimize import linear_sum_assignment
from scipy.spatial.distance import cdist

from . import _lib

_contexts = {}


def _context(index):
    if index not in _contexts:
        _contexts[index] = _lib.Context(index)
    return _contexts[index]


def nn_sq_dist(a, b, return_index=False):
    """a (B,N,3), b (B,M,3) fp32 DEVICE tensors -> (B,N) squared distance of every a-point to its nearest b-point
    (and its index) through the C ABI."""
    if not (a.is_cuda and b.is_cuda and a.device == b.device):
        raise _lib.LsdmError("nn_sq_dist runs in liblsdm_hip.so on device tensors; got %s / %s" % (a.device, b.device))
    a, b = a.float().contiguous(), b.float().contiguous()
    if a.dim() != 3 or b.dim() != 3 or a.shape[0] != b.shape[0] or a.shape[2] != 3 or b.shape[2] != 3:
        raise ValueError("expected (B,N,3) and (B,M,3)")
    d2 = torch.empty(a.shape[:2], device=a.device, dtype=torch.float32)
    idx = torch.empty(a.shape[:2], device=a.device, dtype=torch.int32) if return_index else None
    _context(a.device.index).call("lsdm_nn_distance", a.data_ptr(), a.shape[1], b.data_ptr(), b.shape[1], a.shape[0], d2.data_ptr(),
                                  None if idx is None else idx.data_ptr(), torch.cuda.current_stream(a.device).cuda_stream)
    return (d2, idx) if return_index else d2


def _nn_sq_dist_host(a, b):
    d = a.float()[:, :, None, :] - b.float()[:, None, :, :]
    return (d * d).sum(-1).min(dim=2)[0]


def chamfer_distance(x, y):
    """x (B,N,3), y (B,M,3) -> scalar tensor."""
    nn = nn_sq_dist if x.is_cuda else _nn_sq_dist_host
    return nn(x, y).mean(dim=1).mean() + nn(y, x).mean(dim=1).mean()


def emd(x, y):
    if x.dim() == 3:
        x, y = x.squeeze(0), y.squeeze(0)
    d = cdist(x.detach().cpu().numpy(), y.detach().cpu().numpy())
    r, c = linear_sum_assignment(d)
    return d[r, c].sum() / min(len(x), len(y))


def accuracy(output, target, topk=(1,)):
    maxk = max(topk)
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.view(1, -1).expand_as(pred.t()))
    return [correct[:k].reshape(-1).float().sum(0).mul_(100.0 / target.size(0)) for k in topk]


def calculate_fscore(gt, pr, th=0.1):
    """gt (N,3), pr (M,3) -> (fscore, precision, recall)."""
    if len(gt) == 0 or len(pr) == 0:
        return 0, 0, 0
    nn = nn_sq_dist if gt.is_cuda else _nn_sq_dist_host
    d1 = nn(gt[None], pr[None])[0].sqrt()      # gt -> pr
    d2 = nn(pr[None], gt[None])[0].sqrt()      # pr -> gt
    recall = float((d2 < th).sum()) / len(d2)
    precision = float((d1 < th).sum()) / len(d1)
    f = 2 * recall * precision / (recall + precision) if recall + precision > 0 else 0
    return f, precision, recall
