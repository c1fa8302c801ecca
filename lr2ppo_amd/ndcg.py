"""NDCG with the reference's definition (ndcg.py:9-65): gain 2^rel - 1, discount log2(i + 2),
k in {1, 3, 5, 10, 20, 1e8}, NDCG := 1 when the ideal DCG <= 1e-6.

`ndcg_rows` is what the evaluate() functions of this package use: the scores of a whole split stay on the device and ONE
launch of lr2_ndcg (csrc/misc.hip) produces the [items, 6] table (the reference loops in Python per item and per rank,
13 s per evaluation in its logs).  `AverageNDCGMeter` keeps the reference's host-side class API (vectorised)."""
import torch


def ndcg_rows(scores, golds, device, ks=(1, 3, 5, 10, 20, 100000000)):
    """scores / golds: per-item 1-D tensors (device scores, labels anywhere) -> CPU fp32 [items, len(ks)]."""
    from . import ops
    if not scores:
        return torch.zeros(0, len(ks))
    sizes = [int(s.numel()) for s in scores]
    if max(sizes) > 64:
        raise ValueError("an item carries more than 64 tags (lr2_ndcg's per-item limit)")
    offsets = torch.tensor([0] + sizes, dtype=torch.int64).cumsum(0).to(device)
    flat_s = torch.cat([s.reshape(-1).to(device=device, dtype=torch.float32) for s in scores])
    flat_g = torch.cat([g.reshape(-1).to(device=device, dtype=torch.int64) for g in golds])
    return ops.ndcg(flat_s, flat_g, offsets, tuple(ks)).cpu()


class AverageNDCGMeter(object):
    def __init__(self, ndcg_at_k=(1, 3, 5, 10, 20, 100000000)):
        self.ndcg_at_k = list(ndcg_at_k)
        self.reset()

    def reset(self):
        self.ndcg = {k: [] for k in self.ndcg_at_k}

    def value(self):
        for k in self.ndcg:
            self.ndcg[k] = torch.mean(torch.stack([torch.as_tensor(v, dtype=torch.float32) for v in self.ndcg[k]]))
        return self.ndcg

    @staticmethod
    def _dcg_prefix(relevances):
        rel = relevances.to(torch.int64)
        gains = (2 ** rel - 1).to(torch.float32)
        disc = torch.log2(torch.arange(rel.numel(), dtype=torch.int64) + 2)
        return torch.cumsum(gains / disc, dim=0)

    def compute_dcg_at_k(self, relevances, k):
        n = min(len(relevances), k)
        return self._dcg_prefix(relevances)[n - 1] if n > 0 else torch.zeros(())

    def return_ndcg_at_k(self, predicted_relevance, true_relevances):
        """[len(ks)] NDCG vector for one item given gold labels in predicted order and in ideal order."""
        pred, true = self._dcg_prefix(predicted_relevance.cpu()), self._dcg_prefix(true_relevances.cpu())
        out = []
        for k in self.ndcg_at_k:
            n = min(pred.numel(), k)
            t = true[n - 1]
            out.append(torch.ones(()) if t <= 1e-6 else pred[n - 1] / t)
        return torch.stack(out).to(torch.float32)

    def return_ndcg_at_k_from_scores(self, scores, gold):
        """evaluate()'s per-item recipe (finetune/ppo.py:651-659): sort by score, compare with the ideal order."""
        _, idx = torch.sort(scores, dim=-1, descending=True)
        true_rel, _ = torch.sort(gold, dim=-1, descending=True)
        return self.return_ndcg_at_k(gold[idx], true_rel)

    def compute_ndcg_at_k(self, predicted_relevance, true_relevances):
        vec = self.return_ndcg_at_k(predicted_relevance, true_relevances)
        for i, k in enumerate(self.ndcg_at_k):
            self.ndcg[k].append(vec[i])
