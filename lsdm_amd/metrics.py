"""Evaluation metrics written by the CLI (run/test_sdm.py:185-208), without pytorch3d / open3d.

chamfer   : pytorch3d.loss.chamfer_distance defaults -- mean squared nearest-neighbour distance, both directions
            summed, averaged over the batch (reference call site run/test_sdm.py:186).
emd       : util/evaluation.py:5-11 (Hungarian assignment on the Euclidean distance matrix).
accuracy  : util/evaluation.py:13-26 (precision@k in percent).
f-score   : util/evaluation.py:28-52 (threshold 0.1 on nearest-neighbour Euclidean distances).
These post-process the sample on the host; they are not part of the accelerated path.
"""
import torch
from scipy.optimize import linear_sum_assignment
from scipy.spatial.distance import cdist


def chamfer_distance(x, y):
    d = torch.cdist(x.float(), y.float()) ** 2      # (B,N,M)
    return d.min(dim=2)[0].mean(dim=1).mean() + d.min(dim=1)[0].mean(dim=1).mean()


def emd(x, y):
    if x.dim() == 3:
        x, y = x.squeeze(0), y.squeeze(0)
    d = cdist(x.numpy(), y.numpy())
    r, c = linear_sum_assignment(d)
    return d[r, c].sum() / min(len(x), len(y))


def accuracy(output, target, topk=(1,)):
    maxk = max(topk)
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.view(1, -1).expand_as(pred.t()))
    return [correct[:k].reshape(-1).float().sum(0).mul_(100.0 / target.size(0)) for k in topk]


def calculate_fscore(gt, pr, th=0.1):
    d = torch.cdist(gt.float(), pr.float())
    d1, d2 = d.min(dim=1)[0], d.min(dim=0)[0]       # gt->pr, pr->gt
    if len(d1) == 0 or len(d2) == 0:
        return 0, 0, 0
    recall = float((d2 < th).sum()) / len(d2)
    precision = float((d1 < th).sum()) / len(d1)
    f = 2 * recall * precision / (recall + precision) if recall + precision > 0 else 0
    return f, precision, recall
