"""SPrompts — the continual-learning driver plugin surface of the reference (methods/sprompt.py:104-694) over the HIP
network.  Same constructor argument, same public attributes/methods the trainer uses (``_network``,
``incremental_train()``, ``after_task()``), same hot loop (forward -> cal_loss -> sum -> backward -> SGD step), same
trainable-parameter filter, optimiser and scheduler, same evaluation protocol and ``final_res`` structure.

What changed, and why:
  * device handling is explicit (no hard ``.cuda()``): everything follows ``args['device'][0]``;
  * data parallelism actually works: with torch.distributed initialised (one process per GPU, RCCL), the features are
    all-gathered for the global contrastive matrix and the prompt-factor gradients are SUM all-reduced (lpi_amd/dp.py) —
    the reference's multi-GPU path is dead code (README.md:13);
  * the per-row ``np.argsort`` rank search of ``itm_eval`` (sprompt.py:558-599) runs on the GPU (lpi_retrieval_rank);
  * datasets: ``args['dataset_impl']`` = 'coco' (utils/data.py's Coco / CocoEval on image_root + the two annotation files; default when
    image_root exists) or 'synthetic' (default when image_root is missing: COCO is not available offline; loaders with the same
    item structure);
  * the KMeans task keys (sprompt.py:370-397) are fitted on the GPU (lpi_amd/kmeans.py: scikit-learn's algorithm, the features stay in HBM;
    args['kmeans_impl'] = 'sklearn' keeps the reference's host call);
  * the hot loop runs the step FUSED (SliNet.train_step: forward -> losses -> backward in one call, the loss kernels seed the backward) on batches an
    input pipeline prepared one step ahead (lpi_amd/pipeline.py: pinned staging, H2D on a side stream, tokenisation off the critical path) and never
    synchronises the host with the device inside an epoch except where it prints the loss (every 50 batches; the reference calls loss.item() every
    step, sprompt.py:313).  args['fused_step'] = False / args['prefetch'] = False give the reference-shaped loop (net(...) -> cal_loss -> backward).
"""
import collections
import json
import logging
import os
from datetime import datetime

import numpy as np
import torch
from torch import optim
from torch.utils.data import DataLoader

from lpi_amd import _lib
from lpi_amd.retrieval.loss.loss import ClipLoss
from lpi_amd.retrieval.methods.base import BaseLearner
from lpi_amd.retrieval.models.slinet import SliNet
from lpi_amd.retrieval.utils.data import SyntheticCoco, SyntheticCocoEval


class LossLog:
    """The per-key AverageMeters of the hot loop (sprompt.py:294-323) without a device synchronisation or an arithmetic kernel per step: the step's loss
    tensors are only REFERENCED here and averaged when the line is printed."""

    def __init__(self):
        self.items = collections.defaultdict(list)

    def add(self, losses):
        for k, v in losses.items():
            self.items[k].append(v)

    def flush(self):
        out = {}
        for k, vs in self.items.items():
            vals = [sum(float(p) for p in v) if isinstance(v, (tuple, list)) else float(v) for v in vs]
            out[k] = sum(vals) / max(1, len(vals))
        self.items.clear()
        return out


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def _dist_world():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _dist_rank():
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class SPrompts(BaseLearner):
    def __init__(self, args):
        super().__init__(args)
        if args["net_type"] == "slip":
            self._network = SliNet(args)
        else:
            raise ValueError('Unknown net: {}.'.format(args["net_type"]))       # sprompt.py:118
        self.args = args
        self.EPSILON = args["EPSILON"]
        self.epochs = args["epochs"]
        self.lrate = args["lrate"]
        self.batch_size = args["batch_size"]
        self.weight_decay = args["weight_decay"]
        self.num_workers = args["num_workers"]
        self.num_tasks = int(args.get("num_tasks", 12))
        self.topk = 2
        self.class_num = self._network.class_num
        self.all_keys = []
        self.textual_all_keys = []
        self.loss = ClipLoss()
        self.cur_id = 0
        self.final_res = None

    # ------------------------------------------------------------------ sprompt.py:145-195
    def after_task(self):
        self._old_network = self._network.copy().freeze()
        self._known_classes = self._total_classes
        logging.info('Exemplar size: {}'.format(self.exemplar_size))

    def _datasets(self, i):
        impl = self.args.get("dataset_impl")
        if impl is None:
            impl = "coco" if os.path.isdir(str(self.args.get("image_root", ""))) else "synthetic"
        if impl == "coco":          # sprompt.py:163-170
            from lpi_amd.retrieval.utils.data import Coco, CocoEval
            # pixel_format = 'u8': the datasets hand over the decoded uint8 pixels and ToTensor + Normalize run inside the im2col kernel (lpi_patchify_u8:
            # bit for bit the f32 pipeline, a quarter of the host-to-device bytes)
            pf = self.args.get('pixel_format', 'f32')
            return (Coco(image_root=self.args['image_root'], ann_file=self.args['annotation_train_root'], tasks=[i], pixel_format=pf),
                    CocoEval(image_root=self.args['image_root'], ann_file=self.args['annotation_val_root'], tasks=np.arange(0, i + 1),
                             eval_transform=self.args.get('eval_transform', 'center'), pixel_format=pf))
        if impl != "synthetic":
            raise ValueError(f"unknown dataset_impl {impl!r} (coco | synthetic)")
        res = self._network.clip_cfg.image_resolution
        n_train = int(self.args.get("synthetic_train_size", 4 * self.batch_size))
        n_eval = int(self.args.get("synthetic_eval_images_per_task", 16))
        # synthetic_captions = 'strings': items carry caption STRINGS like the reference's Coco (needs a BPE merge table: lpi_amd.synth_bpe.ensure_vocab
        # supplies a synthetic one where CLIP's is absent); synthetic_image_pool = K: K distinct images, generated once (items are views)
        return (SyntheticCoco(n_train, [i], res, seed=i, captions=self.args.get("synthetic_captions", "ids"),
                              image_pool=int(self.args.get("synthetic_image_pool", 0)), pixel_format=self.args.get("pixel_format", "f32")),
                SyntheticCocoEval(n_eval, np.arange(0, i + 1), 2, res, seed=i))

    def incremental_train(self):
        final_res = {}
        for i in range(self.num_tasks):
            self._cur_task = [i]
            self.cur_id = i
            self._network.update_fc(self._total_classes)
            train_dataset, test_dataset = self._datasets(i)
            sampler = None
            if _dist_world() > 1:
                from torch.utils.data.distributed import DistributedSampler
                sampler = DistributedSampler(train_dataset, shuffle=True, drop_last=True)
            # single-process loading: the images stay a list and the pipeline gathers them straight into pinned memory (one copy); with worker
            # processes the batch arrives stacked in shared memory and the pipeline's gather is its pinning copy (no pin_memory thread needed)
            from lpi_amd.retrieval.utils.data import collate_keep_images
            self.train_loader = DataLoader(train_dataset, batch_size=self.batch_size, shuffle=sampler is None, sampler=sampler,
                                           num_workers=self.num_workers, drop_last=sampler is not None,
                                           collate_fn=collate_keep_images if self.num_workers == 0 else None,
                                           persistent_workers=self.num_workers > 0)
            self.test_loader = DataLoader(test_dataset, batch_size=128, shuffle=False, num_workers=self.num_workers, pin_memory=True)
            final_res[i] = self._train(self.train_loader, self.test_loader)
        self.final_res = final_res
        if _dist_rank() == 0:       # every rank holds the same keys and evaluates the same test set: one writer
            os.makedirs('./res', exist_ok=True)
            # the evaluation-transform choice travels with the numbers (utils/data.py CocoEval: 'center' | 'reference')
            self.save_dict({**final_res, 'eval_transform': self.args.get('eval_transform', 'center')}, f'./res/{datetime.now()}.json')

    def state_dict(self):
        """What a continual run carries from task to task besides the frozen backbone (the reference's BaseLearner.save_checkpoint,
        methods/base.py:57-63, is never called and saves the whole network): the prompt factors, numtask, the KMeans task keys of both modalities."""
        return {"network": self._network.trainable_state_dict(), "cur_id": self.cur_id,
                "all_keys": [k.detach().float().cpu() for k in self.all_keys],
                "textual_all_keys": [k.detach().float().cpu() for k in self.textual_all_keys]}

    def load_state_dict(self, sd):
        self._network.load_trainable_state_dict(sd["network"])
        self.cur_id = int(sd.get("cur_id", self.cur_id))
        self.all_keys = [k.to(self._device) for k in sd["all_keys"]]
        self.textual_all_keys = [k.to(self._device) for k in sd["textual_all_keys"]]
        return self

    def save_checkpoint(self, filename):
        torch.save(self.state_dict(), '{}_{}.pkl'.format(filename, self.cur_id))      # the reference's file name pattern (base.py:63)

    def save_dict(self, dictionary, file_path):
        with open(file_path, 'w') as file:
            json.dump(dictionary, file)

    # ------------------------------------------------------------------ sprompt.py:197-256
    def _train(self, train_loader, test_loader):
        optimizer, scheduler = self._setup_training()
        self.run_epoch = self.epochs
        return self.train_function(train_loader, test_loader, optimizer, scheduler)

    def _setup_training(self):
        """sprompt.py:197-255 up to the hot loop: device, trainable filter, optimiser, scheduler."""
        self._network.to(self._device)
        if _dist_world() > 1 and self._network.exchange is None:
            from lpi_amd.dp import Exchange
            # the modes of the reference's gather_features (sprompt.py:38-82); the default needs no backward collective
            self._network.exchange = Exchange(local_loss=bool(self.args.get("local_loss", False)),
                                              gather_with_grad=bool(self.args.get("gather_with_grad", False)))
        network = self._network
        for name, param in network.named_parameters():
            param.requires_grad_(False)
            if "prompts" + "." + str(network.numtask - 1) + "." in name:           # sprompt.py:235
                param.requires_grad_(True)
        enabled = {n for n, p in network.named_parameters() if p.requires_grad}
        print(f"Parameters to be updated: {enabled}")
        if self.args.get("fused_sgd", True) and str(self._device).startswith("cuda"):
            # the same SGD / cosine arithmetic (sprompt.py:253-255) as one HIP kernel over the task's five factors (lpi_sgd_step).  FlatSGD RE-SEATS the
            # five parameters' .data as views of one flat buffer: the module must already be on its device (it is: .to() above) and must not be moved
            # or cast while this optimiser lives — FlatSGD.step() checks the seating and raises if it was lost
            from lpi_amd.optim import CosineLR, FlatSGD
            optimizer = FlatSGD([p for p in network.parameters() if p.requires_grad], lr=self.lrate, momentum=0.9, weight_decay=self.weight_decay)
            scheduler = CosineLR(optimizer, T_max=self.epochs)
        else:
            optimizer = optim.SGD(network.parameters(), momentum=0.9, lr=self.lrate, weight_decay=self.weight_decay)
            scheduler = optim.lr_scheduler.CosineAnnealingLR(optimizer=optimizer, T_max=self.epochs)
        self.run_epoch = self.epochs
        return optimizer, scheduler

    # ------------------------------------------------------------------ sprompt.py:290-334 (hot loop)
    def _batches(self, train_loader):
        """The epoch's batches as objects with .images / .text on the device.  Default: lpi_amd.pipeline.BatchPipeline (batch i+1 gathered into pinned
        memory, tokenised and copied on a side stream while batch i trains); args['prefetch'] = False: the reference's order — `images.cuda()` and the
        tokenizer inside the step (sprompt.py:301-303)."""
        net = self._network
        if self.args.get("prefetch", True) and torch.device(self._device).type == "cuda":
            from lpi_amd.pipeline import BatchPipeline
            pipe = getattr(self, "_pipeline", None)
            if pipe is None or pipe.loader is not train_loader:
                pipe = self._pipeline = BatchPipeline(train_loader, self._device, net.prepare_text, depth=int(self.args.get("prefetch_depth", 3)),
                                                      threads=int(self.args.get("prefetch_threads", 8)), timing=bool(self.args.get("pipeline_timing", False)))
            return iter(pipe)

        def plain():
            from types import SimpleNamespace
            for item in train_loader:
                images, captions = item[0], item[1]
                if not torch.is_tensor(images):
                    images = torch.stack(list(images))
                yield SimpleNamespace(images=images.to(self._device, non_blocking=True), text=captions if torch.is_tensor(captions) else list(captions))
        return plain()

    def train_function(self, train_loader, test_loader, optimizer, scheduler):
        log = LossLog()
        for epoch in range(self.run_epoch):
            self.train_epoch(train_loader, optimizer, epoch, log)
            scheduler.step()
        self.clustering(dataloader=train_loader)
        _, _, final_res = self._evaluate_retrieval(test_loader)
        return final_res

    def train_epoch(self, train_loader, optimizer, epoch=0, log=None, on_step=None):
        """One epoch of the hot loop (sprompt.py:296-323).  on_step(i, batch, model_out) -> True stops the epoch (bench.py times the loop through it)."""
        log = LossLog() if log is None else log
        net = self._network
        fused = bool(self.args.get("fused_step", True)) and torch.device(self._device).type == "cuda"
        flat_grad, grad_views = getattr(optimizer, "flat_grad", None), getattr(optimizer, "grad_views", None)
        net.train()
        if hasattr(getattr(train_loader, "sampler", None), "set_epoch"):
            train_loader.sampler.set_epoch(epoch)          # DistributedSampler: a new shuffle every epoch
        batches = self._batches(train_loader)
        try:
            for i, batch in enumerate(batches):
                if fused:
                    # forward -> cal_loss -> sum -> backward (sprompt.py:303-310) in one call; data parallel: the all-gather / all-reduce are inside it
                    model_out = net.train_step(batch.images, batch.text, flat_grad=flat_grad, grad_views=grad_views)
                else:
                    image_features, text_features, visual_prompt, textual_prompt = net(batch.images, batch.text)
                    model_out = net.cal_loss(image_features, text_features, visual_prompt, textual_prompt)
                    world = net.exchange.world if net.exchange is not None else 1
                    # data-independent terms are identical on every rank: count them once under the SUM all-reduce
                    bw = net.exchange.loss_weight if net.exchange is not None else 1.0
                    loss = sum(v * bw if k == "base_loss" else v / world for k, v in model_out['loss'].items())
                    optimizer.zero_grad()
                    loss.backward()
                    if net.exchange is not None:
                        net.exchange.allreduce_grads([p for p in net.parameters() if p.requires_grad])
                optimizer.step()
                log.add({k: (v.detach() if torch.is_tensor(v) else v) for k, v in model_out['loss'].items()})
                if i % 50 == 0:
                    info = 'Task {}, Epoch {}/{}, Batch {}, lr {:.4f} =>, '.format(
                        self.cur_id, epoch + 1, self.run_epoch, i, optimizer.param_groups[0]["lr"])
                    for k, v in log.flush().items():         # the one place of the loop where the host waits for the device
                        info += '{} = {:.4f}, '.format(k, v)
                    logging.info(info)
                if on_step is not None and on_step(i, batch, model_out):
                    break
        finally:
            if hasattr(batches, "close"):
                batches.close()          # an early exit (on_step, an exception) stops the pipeline's producer thread

    # ------------------------------------------------------------------ sprompt.py:336-397
    def _task_id(self, feature, all_keys):
        """argmin over tasks of the minimum L1 distance to the task's KMeans centres (sprompt.py:343-350) — lpi_l1_task_id."""
        keys = torch.stack([k.to(device=feature.device, dtype=torch.float32) for k in all_keys]).contiguous()      # [T, C, E]
        T, C, E = keys.shape
        f = feature.detach().float().contiguous()
        sel = torch.empty(f.shape[0], dtype=torch.int32, device=f.device)
        _lib.call("lpi_l1_task_id", f.shape[0], E, T, C, f, E, keys, sel, None, torch.cuda.current_stream().cuda_stream)
        return sel.long()

    def get_visual_task_id(self, inputs):
        with torch.no_grad():
            return self._task_id(self._network.extract_vector(inputs), self.all_keys)

    def get_textual_task_id(self, inputs):
        with torch.no_grad():
            return self._task_id(self._network.extract_textual_vector(inputs), self.textual_all_keys)

    def clustering(self, dataloader):
        """sprompt.py:370-397: the un-prompted features of the task's training set, L2-normalised, KMeans(n_clusters=5, random_state=0) per modality ->
        the task's keys.  args['kmeans_impl'] = 'hip' (default: lpi_amd.kmeans — scikit-learn's algorithm with the features left on the GPU, pinned to the
        reference's centres by tests/golden/kmeans.npz) | 'sklearn' (the reference's own call on the host)."""
        impl = self.args.get("kmeans_impl", "hip")
        if impl not in ("hip", "sklearn"):
            raise ValueError(f"unknown kmeans_impl {impl!r} (hip | sklearn)")
        vf, tf = [], []
        for item in dataloader:
            inputs, captions = item[0], item[1]
            if not torch.is_tensor(inputs):
                inputs = torch.stack(list(inputs))
            with torch.no_grad():
                v = self._network.extract_vector(inputs.to(self._device))
                t = self._network.extract_textual_vector(captions if torch.is_tensor(captions) else list(captions))
            if impl == "sklearn":      # the reference's lines verbatim (sprompt.py:380-388): torch normalisation, host features
                vf.append((v / v.norm(dim=-1, keepdim=True)).cpu())
                tf.append((t / t.norm(dim=-1, keepdim=True)).cpu())
                continue
            s = torch.cuda.current_stream().cuda_stream
            for f, acc in ((v.float().contiguous(), vf), (t.float().contiguous(), tf)):      # v / v.norm(dim=-1, keepdim=True): lpi_l2norm_fwd
                o, inv = torch.empty_like(f), torch.empty(f.shape[0], device=f.device)
                _lib.call("lpi_l2norm_fwd", f.shape[0], f.shape[1], f, f.shape[1], o, f.shape[1], inv, s)
                acc.append(o)
        vf, tf = torch.cat(vf, 0), torch.cat(tf, 0)
        if _dist_world() > 1:       # every rank clusters the features of ALL shards, so the task keys are identical everywhere
            from lpi_amd.dp import all_gather_rows
            vf, tf = all_gather_rows(vf), all_gather_rows(tf)      # one padded all_gather_into_tensor each: no pickling, no host trip on RCCL
        if impl == "hip":
            from lpi_amd.kmeans import kmeans_fit
            vc, _, _ = kmeans_fit(vf, 5, random_state=0)
            tc, _, _ = kmeans_fit(tf, 5, random_state=0)
            self.all_keys.append(vc)
            self.textual_all_keys.append(tc)
            return
        from sklearn.cluster import KMeans
        vc = KMeans(n_clusters=5, random_state=0).fit(vf.numpy())
        tc = KMeans(n_clusters=5, random_state=0).fit(tf.numpy())
        self.all_keys.append(torch.tensor(vc.cluster_centers_).to(self._device))
        self.textual_all_keys.append(torch.tensor(tc.cluster_centers_).to(self._device))

    # ------------------------------------------------------------------ sprompt.py:433-548
    @torch.no_grad()
    def _evaluate_retrieval(self, data_loader):
        self._network.eval()
        ds = data_loader.dataset
        texts, texts_cat = ds.text, torch.tensor(ds.text_cat)
        num_text = len(texts)
        text_bs = 256
        image_feats, category_i = [], []
        for image, img_id, category in data_loader:
            image = image.to(self._device)
            selection = self.get_visual_task_id(image)
            image_feats.append(self._network.visual_interface(image, selection))
            category_i.extend(int(z) for z in category)
        text_feats = []
        for i in range(0, num_text, text_bs):
            text = texts[i: min(num_text, i + text_bs)]
            sel = self.get_textual_task_id(text)
            text_feats.append(self._network.textual_interface(text, sel))
        image_feats, text_feats = torch.cat(image_feats), torch.cat(text_feats)
        from lpi_amd.engine import score_matrix
        score_i2t, score_t2i = score_matrix(image_feats, text_feats)      # sprompt.py:509: (I @ T^T) and its transpose, f32 MFMA GEMM
        final_res = self.itm_eval(score_i2t, score_t2i, ds.txt2img, ds.img2txt, category_i, texts_cat)
        return score_i2t.cpu().numpy(), score_t2i.cpu().numpy(), final_res

    # ------------------------------------------------------------------ sprompt.py:550-646
    @torch.no_grad()
    def itm_eval(self, scores_i2t, scores_t2i, txt2img, img2txt, category_i, category_t):
        dev = self._device if torch.device(self._device).type == "cuda" else "cuda:0"
        s_i2t = torch.as_tensor(scores_i2t, dtype=torch.float32, device=dev).contiguous()
        s_t2i = torch.as_tensor(scores_t2i, dtype=torch.float32, device=dev).contiguous()
        n_img, n_txt = s_i2t.shape
        gmax = max(len(v) for v in img2txt.values())
        gt_i = torch.full((n_img, gmax), -1, dtype=torch.int32)
        for i in range(n_img):
            gt_i[i, :len(img2txt[i])] = torch.tensor(img2txt[i], dtype=torch.int32)
        gt_t = torch.tensor([txt2img[t] for t in range(n_txt)], dtype=torch.int32).view(-1, 1)
        r_i = torch.zeros(n_img, dtype=torch.int32, device=dev)
        r_t = torch.zeros(n_txt, dtype=torch.int32, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        _lib.call("lpi_retrieval_rank", n_img, n_txt, s_i2t, n_txt, gt_i.to(dev), gmax, r_i, s)
        _lib.call("lpi_retrieval_rank", n_txt, n_img, s_t2i, n_img, gt_t.to(dev), 1, r_t, s)
        ranks_i, ranks_t = r_i.cpu().numpy(), r_t.cpu().numpy()
        category_i = np.asarray(category_i)
        category_t = np.asarray(category_t)
        task_num = self.cur_id + 1
        i2t_res, t2i_res = {}, {}
        for task in range(task_num):
            r = ranks_i[category_i == task]
            i2t_res[task] = [100.0 * float((r < k).sum()) / len(r) for k in (1, 5, 10)]
            r = ranks_t[category_t == task]
            t2i_res[task] = [100.0 * float((r < k).sum()) / len(r) for k in (1, 5, 10)]
        final_res = {'mscoco': {'i2t': i2t_res, 't2i': t2i_res}}
        logging.info(final_res)
        return final_res
