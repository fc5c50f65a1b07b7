"""TriCoLoNet container - mirrors /root/reference/tricolo/model/tricolo_net.py:11-158 on the HIP-backed modules.

Same construction from a Hydra-style cfg (encoders chosen by NAME in cfg.model.{text,image,voxel}_encoder and built
from cfg.model.modules.<name> via `_target_`, loss from cfg.loss[cfg.loss.name], optimizer from cfg.optimizer),
same forward / _calculate_losses / training_step / validation_step / test_step hooks and the same loss names
(`train_loss/text_image_loss`, ... `total_loss`), so Lightning's Trainer can drive it when Lightning is installed;
`tricolo_amd.train.fit` drives it when it is not.  Modality order text, image, voxel matters: alpha_weight is
asymmetric and the earlier modality is `zis` (tricolo_net.py:47-53,59-63).
"""
import os
import pickle
import types
from itertools import combinations

import numpy as np
import torch

from .. import config as tcfg
from .. import ops
from ..evaluation.eval_retrieval import compute_metrics
from ..layers import TriModule

try:
    import hydra as _hydra
    _instantiate = _hydra.utils.instantiate
except Exception:                                  # noqa: BLE001  Hydra is not installed on the build / GPU boxes
    _instantiate = tcfg.instantiate


def _load_clip(name, device):
    """CLIP is only consulted for `.visual.output_dim` (clip_text.py:10); weights cannot be downloaded here."""
    try:
        import clip
        model = clip.load(name, device=device)[0]
        for p in model.parameters():
            p.requires_grad = False               # tricolo_net.py:22-24
        return model
    except Exception:                              # noqa: BLE001
        dim = {"ViT-L/14": 768, "ViT-B/32": 512, "ViT-B/16": 512}.get(name, 768)
        return types.SimpleNamespace(visual=types.SimpleNamespace(output_dim=dim), parameters=lambda: [])


class TriCoLoNet(TriModule):
    def __init__(self, cfg):
        super().__init__()
        if hasattr(TriModule, "save_hyperparameters"):
            self.save_hyperparameters()            # Lightning: stores cfg in the checkpoint (tricolo_net.py:14)
        else:
            self.__dict__["_hparams_ns"] = types.SimpleNamespace(cfg=cfg)
        self._cfg = cfg
        self.image_encoder = None
        self.voxel_encoder = None
        clip_model = None
        if cfg.model.image_encoder == "CLIPImageEncoder" or cfg.model.text_encoder == "CLIPTextEncoder":
            clip_model = _load_clip(cfg.model.modules.clip_model, self.device)
        self.text_encoder = _instantiate(getattr(cfg.model.modules, cfg.model.text_encoder), clip_model=clip_model)
        if cfg.model.image_encoder is not None:
            self.image_encoder = _instantiate(getattr(cfg.model.modules, cfg.model.image_encoder), clip_model=clip_model)
        if cfg.model.voxel_encoder is not None:
            self.voxel_encoder = _instantiate(getattr(cfg.model.modules, cfg.model.voxel_encoder))
            # (rounds 2-3 switched SparseCNNEncoder.fuse_pool_reduce off beside an image tower at 32^3 - the step was 1 % slower then;
            #  re-measured in round 4, three alternating pairs: fused 2.757 against 2.768 ms, with the image tower issued first 2.740)
        self.loss_fn = _instantiate(getattr(cfg.loss, cfg.loss.name))
        self.val_test_step_outputs = []
        self.overlap_towers = os.environ.get("TRICOLO_OVERLAP", "1") != "0"
        self.__dict__["dp_split"] = None            # parallel.BackwardSplit (data-parallel gradient overlap): gates the voxel output
        self.__dict__["_side_streams"] = None

    def _apply(self, fn, *args, **kwargs):
        """Every device move (.to / .cuda) also creates ops.one() - the cached loss-gradient scalar - on the parameters' device, eagerly:
        a step first run INSIDE a HIP-graph capture (torch.optim, a custom capture without FusedAdam.prepare()) then finds it."""
        out = super()._apply(fn, *args, **kwargs)
        p = next(self.parameters(), None)
        if p is not None and p.is_cuda and not torch.cuda.is_current_stream_capturing():
            ops.one(p.device)
        return out

    # Lightning supplies .hparams / log_dict / log / print when present; minimal stand-ins otherwise
    def __getattr__(self, name):
        if name == "hparams" and "_hparams_ns" in self.__dict__:
            return self.__dict__["_hparams_ns"]
        return super().__getattr__(name)

    def _log_dict(self, d, **kw):
        fn = getattr(super(), "log_dict", None)
        if fn is not None:
            fn(d, **kw)

    def _log(self, k, v):
        fn = getattr(super(), "log", None)
        if fn is not None:
            fn(k, v)
        else:
            self.__dict__.setdefault("logged", {})[k] = v

    def configure_optimizers(self):
        return _instantiate(self._cfg.optimizer, params=self.parameters())

    def forward(self, data_dict):
        """Towers are independent until the loss (tricolo_net.py:46-54 runs them serially on one stream).  On a GPU the
        text and voxel towers - latency-bound, few workgroups - run on side HIP streams next to the ResNet tower; autograd
        replays each tower's backward on the stream its forward ran on, so the overlap holds both ways and is captured as
        parallel branches of the HIP graph.  Key order (text, image, voxel) is preserved: it fixes which side gets alpha."""
        tokens = data_dict["tokens"]
        n_side = int(self.image_encoder is not None) + int(self.voxel_encoder is not None)
        overlap = tokens.is_cuda and self.overlap_towers and n_side >= 1 and torch.is_grad_enabled()
        if not overlap:
            output_dict = {"text_features": self.text_encoder(tokens, data_dict)}
            if self.image_encoder is not None:
                output_dict["image_features"] = self.image_encoder(data_dict["images"].flatten(end_dim=1), data_dict)
            if self.voxel_encoder is not None:
                output_dict["voxel_features"] = self._gate(self.voxel_encoder(data_dict["voxels"], len(data_dict["model_id"])))
            return output_dict
        main = torch.cuda.current_stream()
        ops.stamp("step.start")
        if self._side_streams is None:
            self._side_streams = (torch.cuda.Stream(), torch.cuda.Stream())
        s_text, s_vox = self._side_streams
        # (round 4: both side towers on ONE stream, text first, was measured too: the image forward ends at 0.84 ms instead of 0.94 -
        #  nothing runs beside its first half - but the side stream's kernels queue behind the image tower's down-sample branch until
        #  ~0.65 ms, the voxel forward ends at 1.17 ms and the step takes 2.87 ms against 2.64)
        # The heaviest tower runs on the caller's stream: the image tower, or - Bi(V) - the voxel tower; the others on side streams
        # forked at the start of the step.  ISSUE order matters under HIP-graph replay: the executor folds capture streams onto a
        # few internal streams and a branch issued later queues behind branches issued earlier - including another tower's side
        # branches (tools/step_timeline.py shows it).  Image tower first: autograd, which runs the latest-created node first, then
        # issues the voxel and text backward BEFORE the image tower's and both start right after the loss; the price is a text
        # forward that starts late (it still ends before the loss needs it, +0.05 ms).  Side towers first ("tvi") starts every forward
        # at once but queues the voxel backward behind the image tower's down-sample branches: it starts 1.2 ms late and becomes
        # the last kernel of the step.  Six orders measured: itv 3.34-3.44 ms, tvi 3.42-3.45, vti 3.42, vit 3.49, ivt 3.54, tiv 3.60.
        # (Issuing the side towers from inside the image tower's forward, right after its stem, was tried too: the text forward
        # still starts late - the folding is not a pure function of the issue order - and the step time is the same.)
        # Eager steps are bound by the host's launch rate instead: there the short towers go first so that their kernels are
        # already queued on their streams while the host spends ~1.5 ms issuing the image tower (6.1 against 7.6 ms per step).
        # Re-measured with the row-unit conv kernels (shorter layer1 / layer4, bench.py three times each): trimodal tvi 3.30 /
        # itv 3.33-3.35 / vti 3.33-3.35 ms (config 5: 24.21 / 24.20); without a voxel tower (config 3) itv 4.67 / tvi 4.73.
        # Round 4 (krow weight gradients, shorter side towers; three alternating runs each): itv 2.756 / tvi 2.768 ms, with the fused
        # voxel pool routing 2.740 / 2.757 - the image tower is issued first in every configuration now.
        # Round 5, Bi(V) (no image tower, voxel tower on the caller's stream): the voxel forward is the critical chain there and, issued
        # after the text tower, its first kernel queued 34 us behind the text tower's first four - "vt" 0.881 / 0.890 ms against 0.895 / 0.903.
        capt_order = "itv" if self.image_encoder is not None else "vt"
        order = os.environ.get("TRICOLO_TOWER_ORDER") or (capt_order if torch.cuda.is_current_stream_capturing() else "tvi")
        s_text.wait_stream(main)
        vox = img = text = None
        vox_on_main = self.image_encoder is None
        if self.voxel_encoder is not None and not vox_on_main:
            s_vox.wait_stream(main)
        out = {}

        def run_tower(which):
            if which == "t":
                with torch.cuda.stream(s_text):
                    out["t"] = self.text_encoder(tokens, data_dict)
                    ops.stamp("text.fwd.end")
            elif which == "v" and self.voxel_encoder is not None:
                with torch.cuda.stream(main if vox_on_main else s_vox):
                    out["v"] = self._gate(self.voxel_encoder(data_dict["voxels"], len(data_dict["model_id"])))
                    ops.stamp("voxel.fwd.end")
        # (round 6: issuing the side towers from INSIDE the image tower's forward - behind its stem / max-pool / layer1 / layer2 / layer3, depending
        #  on the start of the step only - and lending the text stream to the image tower's shortcut branches were both measured again on the new
        #  issue orders: within +-0.5 % of this order on two boxes; the hooks were removed)
        # (round 6, measured and removed: creating the image tower's autograd node BEHIND the side towers' - forward kernels still issued first -
        #  so that autograd issues its backward, the critical chain, before theirs: the voxel backward then starts at 1.59 ms instead of 0.93 and
        #  the step is the same, 2.486 against 2.477-2.493 ms)
        for which in order:
            if which == "i" and self.image_encoder is not None:
                img = self.image_encoder(data_dict["images"].flatten(end_dim=1), data_dict)
                ops.stamp("image.fwd.end")
            else:
                run_tower(which)
        text, vox = out.get("t"), out.get("v")
        main.wait_stream(s_text)
        text.record_stream(main)
        output_dict = {"text_features": text}
        if img is not None:
            output_dict["image_features"] = img
        if vox is not None:
            if self.image_encoder is not None:
                main.wait_stream(s_vox)
                vox.record_stream(main)
            output_dict["voxel_features"] = vox
        return output_dict

    def join_side_streams(self):
        """The current stream waits for everything issued on the towers' side streams.  Call between backward() and the optimizer step
        (parallel.dp_training_step does): autograd's own end-of-backward synchronisation covers the streams its AccumulateGrad nodes ran
        on, which need not be the streams the towers' backward kernels ran on once those nodes outlive an iteration (torch warns about
        exactly that) - round 6 found the text tower's parameters ending a replayed step on other bits once in ~300 replays."""
        if not torch.cuda.is_available():
            return
        cur = torch.cuda.current_stream()
        streams = list(self._side_streams or ())
        for enc in (self.image_encoder, self.voxel_encoder, self.text_encoder):
            for name in ("_side", "_side_ds", "_side_prep"):
                side = getattr(enc, name, None) if enc is not None else None
                if side is not None and getattr(side, "stream", None) is not None:
                    streams.append(side.stream)
        for st in streams:
            if st is not None and st != cur:
                cur.wait_stream(st)

    def _gate(self, z):
        return self.dp_split.gate(z) if (self.dp_split is not None and torch.is_grad_enabled() and z.requires_grad) else z

    def _calculate_losses(self, output_dict, loss_prefix):
        loss_dict = {}
        fused = getattr(self.loss_fn, "all_pairs", None)
        res = fused(list(output_dict.values())) if (fused is not None and len(output_dict) in (2, 3)) else None
        if res is not None:                        # same names, same values, same summation order as the loop below - one autograd node
            for (a, b), l in zip(combinations(output_dict.keys(), 2), res[0]):
                loss_dict[f"{loss_prefix}/{a[:-9]}_{b[:-9]}_loss"] = l
            loss_dict[f"{loss_prefix}/total_loss"] = res[1]
            ops.stamp("loss.fwd.end")
            return loss_dict
        for a, b in combinations(output_dict.keys(), 2):
            loss_dict[f"{loss_prefix}/{a[:-9]}_{b[:-9]}_loss"] = self.loss_fn(output_dict[a], output_dict[b])
        loss_dict[f"{loss_prefix}/total_loss"] = sum(loss_dict.values())
        return loss_dict

    def training_step(self, data_dict, idx=0):
        output_dict = self(data_dict)
        loss_dict = self._calculate_losses(output_dict, "train_loss")
        self._log_dict(loss_dict, on_step=True, on_epoch=False)
        return loss_dict["train_loss/total_loss"]

    def _stash(self, data_dict, output_dict):
        out = {k: v.detach().cpu().numpy() for k, v in output_dict.items()}
        reduced = {"model_id": data_dict["model_id"], "category": data_dict["category"],
                   "tokens": data_dict["tokens"].cpu().numpy()}
        self.val_test_step_outputs.append((reduced, out))

    def validation_step(self, data_dict, idx=0):
        output_dict = self(data_dict)
        loss_dict = self._calculate_losses(output_dict, "val_loss")
        self._log_dict(loss_dict, on_step=True, on_epoch=False)
        self._stash(data_dict, output_dict)

    def on_validation_epoch_end(self):
        embeddings_dict = self._collate_output()
        self.val_test_step_outputs.clear()
        pr_at_k = compute_metrics(self._cfg.data.dataset, embeddings_dict)
        self._log("val_eval/RR@1", pr_at_k["recall_rate"][0] * 100)
        self._log("val_eval/RR@5", pr_at_k["recall_rate"][4] * 100)
        self._log("val_eval/NDCG@5", pr_at_k["ndcg"][4] * 100)
        self._log("val_eval/MRR", pr_at_k["mrr"] * 100)
        return pr_at_k

    def test_step(self, data_dict, idx=0):
        self._stash(data_dict, self(data_dict))

    def on_test_epoch_end(self):
        embeddings_dict = self._collate_output()
        self.val_test_step_outputs.clear()
        pr = None
        if self._cfg.inference.evaluate:
            pr = compute_metrics(self._cfg.data.dataset, embeddings_dict, print_results=True)
        if self._cfg.inference.save_predictions:
            os.makedirs(self._cfg.inference.output_dir, exist_ok=True)
            path = os.path.join(self._cfg.inference.output_dir, "output.p")
            with open(path, "wb") as f:
                pickle.dump(embeddings_dict, f)
            print(f"\nPredictions saved at {path}")
        return pr

    def _collate_output(self):
        """tricolo_net.py:125-158: shape embedding = image + voxel features, summed and NOT re-normalised."""
        text, shape, model_ids, cats = [], [], [], []
        for data_dict, output_dict in self.val_test_step_outputs:
            text.append(output_dict["text_features"])
            s = np.zeros_like(output_dict["text_features"])
            if "image_features" in output_dict:
                s += output_dict["image_features"]
            if "voxel_features" in output_dict:
                s += output_dict["voxel_features"]
            shape.append(s)
            model_ids.extend(data_dict["model_id"])
            cats.extend(data_dict["category"])
        text, shape = np.vstack(text), np.vstack(shape)
        return {"caption_embedding_tuples": [(None, cats[i], model_ids[i], text[i], shape[i]) for i in range(text.shape[0])]}
