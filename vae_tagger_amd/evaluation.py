"""Batched evaluation caller of the encode+tag path: the counterpart of the reference's evaluation.py
(MultiLabelEvaluator :13-170, evaluate_model :173-200, find_optimal_threshold :202-275) -- SURVEY.md section 8(f) item 4.

Same call surface (`model.encode(pixel_values)` -> `decoder(latents)` -> sigmoid -> threshold, same metric keys and
output files); different mechanics: per batch only the probabilities leave the GPU (one copy), and every metric is
computed once at the end from vectorised numpy -- confusion counts for precision / recall / F1, a rank-based average
precision -- instead of per-class scikit-learn calls.  tests/test_host.py checks the numbers against scikit-learn.
"""
import json
import os

import numpy as np
import torch


def _average_precision(y_true, y_prob):
    """Per-column AP = sum_k (R_k - R_{k-1}) P_k over the distinct score thresholds (scikit-learn's definition);
    columns without a positive get nan."""
    n, c = y_true.shape
    ap = np.full(c, np.nan)
    for j in range(c):
        t = y_true[:, j] > 0
        npos = int(t.sum())
        if npos == 0:
            continue
        order = np.argsort(-y_prob[:, j], kind="stable")
        s, t = y_prob[order, j], t[order]
        last = np.r_[s[1:] != s[:-1], True]              # last element of every run of tied scores
        tp = np.cumsum(t)[last]
        k = (np.nonzero(last)[0] + 1).astype(np.float64)
        recall = tp / npos
        ap[j] = float(np.sum(np.diff(np.r_[0.0, recall]) * (tp / k)))
    return ap


def _prf(tp, fp, fn):
    with np.errstate(divide="ignore", invalid="ignore"):
        p = np.where(tp + fp > 0, tp / (tp + fp), 0.0)
        r = np.where(tp + fn > 0, tp / (tp + fn), 0.0)
        f = np.where(2 * tp + fp + fn > 0, 2 * tp / (2 * tp + fp + fn), 0.0)
    return p, r, f


class MultiLabelEvaluator:
    def __init__(self, class_names=None, device="cuda"):
        self.class_names = class_names
        self.device = device
        self.reset_metrics()

    def reset_metrics(self):
        self.all_predictions, self.all_targets, self.all_probabilities = [], [], []

    @staticmethod
    def _np(x):
        return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)

    def update(self, predictions, targets, probabilities=None):
        """predictions / targets / probabilities: (batch, num_classes) tensors or arrays."""
        self.all_predictions.append(self._np(predictions))
        self.all_targets.append(self._np(targets))
        if probabilities is not None:
            self.all_probabilities.append(self._np(probabilities))

    def compute_metrics(self, threshold=0.5):
        if not self.all_targets:
            raise ValueError("no data: call update() first")
        y_true = np.vstack(self.all_targets) > 0
        y_pred = np.vstack(self.all_predictions) > 0
        y_prob = np.vstack(self.all_probabilities) if self.all_probabilities else y_pred.astype(np.float64)
        n, c = y_true.shape
        tp = (y_true & y_pred).sum(0).astype(np.float64)
        fp = (~y_true & y_pred).sum(0).astype(np.float64)
        fn = (y_true & ~y_pred).sum(0).astype(np.float64)
        support = y_true.sum(0).astype(np.float64)
        p, r, f = _prf(tp, fp, fn)
        m = {"accuracy": float((y_true == y_pred).all(1).mean()), "hamming_loss": float((y_true != y_pred).mean())}
        mp, mr, mf = _prf(tp.sum(), fp.sum(), fn.sum())
        w = support / support.sum() if support.sum() > 0 else np.zeros(c)
        for name, per, micro in (("precision", p, mp), ("recall", r, mr), ("f1", f, mf)):
            m[f"{name}_micro"] = float(micro)
            m[f"{name}_macro"] = float(per.mean())
            m[f"{name}_weighted"] = float((per * w).sum())
        # scikit-learn (what the reference calls, evaluation.py:65-67) scores a class without a positive sample as AP 0 (with a
        # warning) and still averages: macro over all classes, weighted by support, micro over the flattened matrix
        ap = np.nan_to_num(_average_precision(y_true, y_prob), nan=0.0)
        m["mAP"] = float(ap.mean())
        m["mAP_micro"] = float(np.nan_to_num(_average_precision(y_true.reshape(-1, 1), y_prob.reshape(-1, 1)), nan=0.0)[0])
        m["mAP_weighted"] = float((ap * w).sum())
        per_class = {}
        for i in range(c):
            name = self.class_names[i] if self.class_names else f"Class_{i}"
            if support[i] == 0:
                per_class[name] = {"precision": 0.0, "recall": 0.0, "f1": 0.0, "ap": 0.0, "support": 0}
            elif support[i] == n:                        # every sample positive: AP is 1 by convention
                q = float(y_pred[:, i].mean())
                per_class[name] = {"precision": q, "recall": 1.0, "f1": 2 * q / (1 + q) if q > 0 else 0.0, "ap": 1.0,
                                   "support": int(support[i])}
            else:
                per_class[name] = {"precision": float(p[i]), "recall": float(r[i]), "f1": float(f[i]),
                                   "ap": float(ap[i]), "support": int(support[i])}
        m["per_class"] = per_class
        return m

    def print_metrics(self, metrics, detailed=True):
        print(f"subset accuracy {metrics['accuracy']:.4f}   hamming loss {metrics['hamming_loss']:.4f}")
        for k in ("precision", "recall", "f1"):
            print(f"  {k:9s} micro {metrics[k + '_micro']:.4f}  macro {metrics[k + '_macro']:.4f}  weighted {metrics[k + '_weighted']:.4f}")
        print(f"  mAP       macro {metrics['mAP']:.4f}  micro {metrics['mAP_micro']:.4f}  weighted {metrics['mAP_weighted']:.4f}")
        if detailed and "per_class" in metrics:
            print(f"{'':<20} {'Precision':<10} {'Recall':<10} {'F1':<10} {'AP':<10} {'Support':<10}")
            for name, v in metrics["per_class"].items():
                print(f"{name:<20} {v['precision']:<10.4f} {v['recall']:<10.4f} {v['f1']:<10.4f} {v['ap']:<10.4f} {v['support']:<10}")

    def save_metrics(self, metrics, output_path):
        """`<name>_overall.json` + a per-class CSV, as the reference writes them."""
        with open(output_path.replace(".csv", "_overall.json"), "w", encoding="utf-8") as fh:
            json.dump({k: v for k, v in metrics.items() if k != "per_class"}, fh, indent=2, ensure_ascii=False)
        if "per_class" in metrics:
            with open(output_path, "w", encoding="utf-8") as fh:
                fh.write("class_name,precision,recall,f1,ap,support\n")
                for name, v in metrics["per_class"].items():
                    fh.write(f"{name},{v['precision']},{v['recall']},{v['f1']},{v['ap']},{v['support']}\n")


def _check_status(model):
    """After the batch's host copy (a synchronisation point anyway): the encoder's sticky health word (vt_status)."""
    vae = getattr(model, "vae", model)
    ctx = vae._context() if hasattr(vae, "_context") else None
    st = vae.status() if ctx is not None else 0     # on torch's current stream of the model's device: ordered after the encode
    if st & 1:
        raise FloatingPointError("non-finite activations in the encoder: the fp16 residual-stream storage overflowed "
                                 "(vt_set_flag(ctx, 4, 0) stores it as fp32) or the checkpoint holds inf / NaN")
    if st & 2:
        raise FloatingPointError("fp8 mode: activations exceeded the e4m3 range and were clamped (vt_set_flag(ctx, 11, 0) returns to bf16)")


def _probabilities(model, decoder, loader, device):
    """The batched hot path: encode -> decoder -> sigmoid on the GPU; one host copy of the probabilities per batch."""
    probs, labels = [], []
    # this loop reads the health word itself, after the batch's host copy: the wrapper's own per-encode check (a stream synchronise and a
    # 4-byte copy between the encoder's and the decoder's launches, and a read that clears the word before ours) is switched off meanwhile
    had_check = getattr(model, "check_finite", None)
    if had_check is not None:
        model.check_finite = False
    try:
        with torch.no_grad():
            for batch in loader:
                lat = model.encode(batch["pixel_values"].to(device))
                probs.append(torch.sigmoid(decoder(lat)).cpu().numpy())
                labels.append(MultiLabelEvaluator._np(batch["labels"]))
                _check_status(model)
    finally:
        if had_check is not None:
            model.check_finite = had_check
    return np.vstack(probs), np.vstack(labels)


def evaluate_model(model, decoder, test_loader, class_names, device="cuda", threshold=0.5, output_dir=None):
    model.eval(); decoder.eval()
    y_prob, y_true = _probabilities(model, decoder, test_loader, device)
    ev = MultiLabelEvaluator(class_names, device)
    ev.update((y_prob > threshold).astype(np.float32), y_true, y_prob)
    metrics = ev.compute_metrics(threshold)
    ev.print_metrics(metrics)
    if output_dir:
        os.makedirs(output_dir, exist_ok=True)
        ev.save_metrics(metrics, os.path.join(output_dir, "evaluation_results.csv"))
    return metrics


def find_optimal_threshold(model, decoder, val_loader, class_names, device="cuda", output_dir=None):
    """Per-class and global (macro-F1) threshold search over 0.10, 0.15 ... 0.85, all classes at once per threshold."""
    model.eval(); decoder.eval()
    y_prob, y_true = _probabilities(model, decoder, val_loader, device)
    y_true = y_true > 0
    thresholds = np.arange(0.1, 0.9, 0.05)
    c = y_true.shape[1]
    best_f, best_t = np.zeros(c), np.full(c, 0.5)
    g_f, g_t = 0.0, 0.5
    for t in thresholds:
        y_pred = y_prob > t
        tp = (y_true & y_pred).sum(0).astype(np.float64)
        fp = (~y_true & y_pred).sum(0).astype(np.float64)
        fn = (y_true & ~y_pred).sum(0).astype(np.float64)
        f = _prf(tp, fp, fn)[2]
        better = (f > best_f) & (y_true.sum(0) > 0)
        best_f[better], best_t[better] = f[better], t
        if f.mean() > g_f:
            g_f, g_t = float(f.mean()), float(t)
    results = {"global_threshold": g_t, "global_f1": g_f,
               "per_class_thresholds": {n: {"threshold": float(best_t[i]), "f1_score": float(best_f[i])}
                                        for i, n in enumerate(class_names)}}
    print(f"global threshold {g_t:.3f} (macro F1 {g_f:.4f})")
    if output_dir:
        os.makedirs(output_dir, exist_ok=True)
        with open(os.path.join(output_dir, "optimal_thresholds.json"), "w", encoding="utf-8") as fh:
            json.dump(results, fh, indent=2, ensure_ascii=False)
    return results
