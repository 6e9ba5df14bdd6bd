"""Train entrypoint of the MI355X build: ``get_parser()`` / ``train(opt, **kwargs)``.

Mirrors the *interface* of the reference's esrgan.py (flags and their defaults: esrgan.py:30-149 +
options/default.json; step semantics: esrgan.py:399-626; checkpoint and info.json names: esrgan.py:163,370-391;
NaN guard: esrgan.py:645-648) for the hot-path options, so existing launch scripts and hyper-search drivers keep
working.  The iteration itself is ``train.Stepper`` (HIP kernels, optional data parallelism), including the physics loss
heads ``--lambda_nnz/mask/hit/hist`` (fused kernels, losses.py), ``--conditional`` and ``--discriminator standard``.  Outside
this build's scope and therefore raising instead of silently training something else: ``--lambda_wasser``,
the Wasserstein critics, validation/evaluation plots and the HDF5 / text event readers (the jet datasets take ``.npy`` row
tables).

Multi-GPU: launch one process per GPU, e.g.
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m super-resolution_amd.esrgan ...
(the hyphenated package name needs ``python -c "import runpy; runpy.run_module('super-resolution_amd.esrgan', run_name='__main__')"``
or tools/train.py); each rank trains on its shard of every batch and gradients are averaged over RCCL.
"""
import argparse
import json
import math
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import train as _train

# Every key of the reference's options/default.json with its value there (esrgan.py:30-133 declares one flag per key), plus
# the parser-only flags of esrgan.py:117-131 and this build's two additions at the end.
DEFAULTS = dict(
    n_epochs=50, dataset_path="../data/tops_80.h5", dataset_type="spjet", batch_size=8, factor=2, pre_factor=1, lr=0.0002, lr_g=0.0,
    lr_d=0.0, l2decay=0.0, b1=0.9, b2=0.999, n_cpu=0, hr_height=80, hr_width=80, channels=1, scaling_power=1, sample_interval=500,
    image_path="images", checkpoint_interval=500, validation_interval=1000, evaluation_interval=1000, residual_blocks=10,
    warmup_batches=500, pixel_multiplier=1, lambda_pix=0.2, lambda_hr=1, lambda_adv=0.01, lambda_lr=0.1, lambda_hist=0,
    lambda_wasser=0, lambda_nnz=0, lambda_mask=0, lambda_pow=1, lambda_hit=0, hit_threshold=0.5, batchwise_hist=False,
    learn_warmup=True, sigma=500, bins=10, name="", root="", model_path="saved_models", validation_path=None, testset_path=None,
    load_checkpoint=None, report_freq=10, discriminator="patch", relativistic=True, save=True, save_info=None, validate=None,
    n_checkpoints=-1, n_batches=-1, n_validations=-1, n_evaluation=-1, plot_grad=False, smart_save=False, N=5000, wait=[],
    d_threshold=0.001, sinkhorn_eps=0.1, d_channels=[16, 32, 32, 64], n_hardest=None, E_thres=None, set_seed=-1, deterministic=False,
    eval_modes=["E_1", "E_2", "E_3", "E_4", "E_5", "E_10", "E_20", "E_30", "meanimg"], drop_rate=0, res_scale=0.1, lambda_reg=0.01,
    emd_save=False, update_d=1, update_g=1, conditional=False, noise_factor=None, second_discr_reset_interval=0, uniform_init=False,
    uniform_reset=False, save_late=-1,
    # parser-only flags (esrgan.py:117-131)
    use_transposed_conv=False, fully_transposed_conv=False, num_final_res_blocks=0, wasserstein=-1, set_zero_def=[], set_zero_pow=[],
    nth_jet_eval_mode="hr", split_eval=False,
    # this build: synthetic jet batches when no dataset file is at hand (--dataset_type synthetic)
    synthetic_batches=100,
)
# Accepted and without effect here: they steer the reference's validation / evaluation / plotting scaffolding and its data
# loader threads (outside the hot path), so launch scripts and json option files that carry them keep working.
IGNORED = ("n_cpu", "sample_interval", "image_path", "validation_interval", "evaluation_interval", "validation_path", "testset_path",
           "validate", "n_validations", "n_evaluation", "plot_grad", "N", "eval_modes", "emd_save", "deterministic", "sinkhorn_eps",
           "nth_jet_eval_mode", "split_eval", "batchwise_hist")
# Options whose non-default value asks for something this build does not implement: raise instead of training something else.
UNSUPPORTED_POSITIVE = ("lambda_wasser", "second_discr_reset_interval")
UNSUPPORTED_NONDEFAULT = ("smart_save", "wait", "set_zero_def", "set_zero_pow", "uniform_reset")
# flags the reference declares with type=float although their default.json value is an integer literal (esrgan.py:58-120)
FLOAT_FLAGS = {"lr", "lr_g", "lr_d", "l2decay", "b1", "b2", "scaling_power", "pixel_multiplier", "lambda_pix", "lambda_hr", "lambda_adv",
               "lambda_lr", "lambda_hist", "lambda_wasser", "lambda_nnz", "lambda_mask", "lambda_pow", "lambda_hit", "d_threshold",
               "drop_rate", "res_scale", "lambda_reg", "wasserstein", "sigma", "hit_threshold", "E_thres", "noise_factor", "sinkhorn_eps"}
INT_NONE_FLAGS = {"n_hardest"}
STR_LIST_FLAGS = {"wait", "eval_modes", "set_zero_def", "set_zero_pow"}
# the reference's loss_dict keys, in its order (esrgan.py:355-356)
LOSS_KEYS = ['d_loss_def', 'd_loss_pow', 'g_loss', 'def_loss', 'pow_loss', 'adv_loss', 'adv_loss_pow', 'pixel_loss',
             'pixel_loss_pow', 'lr_loss', 'lr_loss_pow', 'hist_loss', 'hist_loss_pow', 'nnz_loss', 'nnz_loss_pow', 'mask_loss',
             'mask_loss_pow', 'wasser_loss', 'wasser_loss_pow', 'hit_loss', 'hit_loss_pow', 'wasser_dist', 'wasser_dist_pow']


def _str2bool(v):
    return v if isinstance(v, bool) else str(v).lower() in ("1", "true", "yes", "y")


def get_parser(argv=None):
    """argparse Namespace with the reference's flag names (esrgan.py:30-133); ``--default file.json`` overlays a json of
    options (or an info.json, whose options sit under "argument"), explicit command-line flags win (esrgan.py:135-147)."""
    ap = argparse.ArgumentParser(description="ESRGAN training on MI355X (super-resolution_amd)")
    for k, v in DEFAULTS.items():
        flag = "-N" if k == "N" else "--" + k          # esrgan.py:86: "-N"
        if k in STR_LIST_FLAGS:
            ap.add_argument(flag, type=str, nargs="+", default=v)
        elif k in INT_NONE_FLAGS:
            ap.add_argument(flag, type=int, default=v)
        elif isinstance(v, bool) or k in ("save_info", "validate"):
            ap.add_argument(flag, type=_str2bool, default=v)
        elif isinstance(v, list):
            ap.add_argument(flag, type=int, nargs="+", default=v)
        elif v is None:
            ap.add_argument(flag, default=None, type=(float if k in FLOAT_FLAGS else str))
        else:
            ap.add_argument(flag, type=(float if k in FLOAT_FLAGS else type(v)), default=v)
    ap.add_argument("--default", type=str, default=None, help="json file with option overrides")
    opt = ap.parse_args(argv)
    if opt.default:
        with open(opt.default) as f:
            over = json.load(f)
        if "argument" in over:                          # an info.json was given (esrgan.py:139-140)
            over = over["argument"]
        for k, v in over.items():
            if k not in DEFAULTS:
                setattr(opt, k, v)                      # keys of other tools (hyper_search's n_histograms, ...) ride along
            elif getattr(opt, k) == DEFAULTS[k]:        # non-default CLI values win
                setattr(opt, k, v)
    return opt


def options(**kw):
    """Programmatic equivalent of get_parser(): defaults + overrides (hyper-search style callers).  Unknown keys are kept as
    attributes (hyper_search.py adds bookkeeping keys of its own to the namedtuple it passes to train())."""
    d = dict(DEFAULTS)
    d.update(kw)
    return SimpleNamespace(**d)


def _opt_dict(opt):
    """esrgan.py:164-167: ``opt`` is an argparse Namespace or, from hyper_search.py:104,159 and the --default path, a namedtuple."""
    try:
        return dict(opt._asdict())
    except AttributeError:
        return dict(vars(opt))


# keys other tools of the reference add to the option set they pass around (hyper_search.py:104,159; info.json bookkeeping):
# carried along silently
FOREIGN_OK = ("n_histograms", "hyper_search", "seed", "default", "name_", "gpu", "synthetic_batches", "dataset", "precision")
_warned = set()


def _warn_once(msg):
    rank = int(os.environ.get("RANK", "0"))
    if rank == 0 and msg not in _warned:
        _warned.add(msg)
        print("[super-resolution_amd] warning: " + msg, flush=True)


def _complete(opt):
    """Namespace carrying every option this module reads: the caller's values over DEFAULTS (a namedtuple built from a json of
    a few options, as constant_args.json + hyper_search.py produce, lacks the rest).  Keys that are neither options of the
    reference nor known bookkeeping keys are kept but reported once (a misspelt `lamda_hist` would otherwise train with the
    default, silently), and so are options of the IGNORED list that were given a non-default value."""
    d = dict(DEFAULTS)
    given = _opt_dict(opt)
    for k, v in given.items():
        if k not in DEFAULTS and k not in FOREIGN_OK:
            _warn_once(f"unknown option '{k}' is carried along but has no effect here (misspelt?)")
        elif k in IGNORED and v != DEFAULTS[k]:
            _warn_once(f"option '{k}' = {v!r} is accepted for compatibility and ignored (it steers the reference's validation / "
                       "evaluation / plotting scaffolding, outside this build's scope)")
    d.update(given)
    return SimpleNamespace(**d)


class SyntheticJets(torch.utils.data.Dataset):
    """Sparse non-negative jet-like images; LR = SumPool(HR) as datasets.py:227,247 builds it."""

    def __init__(self, n, channels, hr_h, hr_w, factor, seed=1234):
        g = torch.Generator().manual_seed(seed)
        self.hr = 10.0 * torch.rand(n, channels, hr_h, hr_w, generator=g) * (torch.rand(n, channels, hr_h, hr_w, generator=g) < 0.1).float()
        self.lr = (factor * factor) * torch.nn.functional.avg_pool2d(self.hr, factor)

    def __len__(self):
        return self.hr.shape[0]

    def __getitem__(self, i):
        return {"lr": self.lr[i], "hr": self.hr[i]}


def _check_supported(opt):
    for k in UNSUPPORTED_POSITIVE:
        if getattr(opt, k, 0) and getattr(opt, k) > 0:
            raise NotImplementedError(f"option {k} > 0 is outside the hot path implemented by this build (SURVEY.md 2.1)")
    for k in UNSUPPORTED_NONDEFAULT:
        if getattr(opt, k, DEFAULTS[k]) != DEFAULTS[k]:
            raise NotImplementedError(f"option {k}={getattr(opt, k)!r} is outside the hot path implemented by this build (SURVEY.md 2.1)")
    if opt.discriminator not in ("patch", "standard") or opt.wasserstein > 0:
        raise NotImplementedError("only the patch (Markovian), standard and conditional discriminators with the relativistic loss are implemented")


def _set_binedges(st, opt, nnz, info):
    """esrgan.py:440-456: bin edges of the energy histogram from the warm-up batches' non-zero pixels: cut at the value
    below which 90 % of them lie, k-means cluster centres (sklearn, random_state=0) as bin centres.  Host-side, once."""
    from sklearn.cluster import KMeans
    nnz = np.array(nnz)
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        # every rank saw only its shard of the warm-up batches: pool the values (rank order) so that all replicas train with
        # the SAME bin edges, the ones a single process on the whole batches would have found and the ones info.json records
        parts = [None] * torch.distributed.get_world_size()
        torch.distributed.all_gather_object(parts, nnz)
        nnz = np.concatenate(parts)
    for k, power in enumerate((1, opt.scaling_power)):
        if st.lambdas[k] <= 0:
            continue
        # upper edge: the left edge of the first of 100 equal bins at which the cumulative count passes 90 % of the values
        values = nnz ** power
        counts, bin_left = np.histogram(values, 100)
        e_max = bin_left[np.argmax(np.cumsum(counts) > 0.9 * len(values))]
        # bin centres: k-means (sklearn, random_state 0) over the un-powered values below e_max, as the reference clusters them;
        # inner edges halfway between neighbouring centres (same arithmetic as the reference: the edges go into info.json)
        below = np.sort(nnz)
        below = below[below <= e_max]
        centres = np.sort(KMeans(n_clusters=opt.bins, random_state=0).fit(below.reshape(-1, 1)).cluster_centers_.ravel())
        edges = np.concatenate(([0.0], centres[:-1] + np.diff(centres) / 2, [e_max]))
        info['binedges%i' % k] = list(edges)
        st.set_hist_binedges(k, edges)


def train(opt, **kwargs):
    """Runs the training loop; returns the ``info`` dict that is also written to ``<model_path>/<name_>info.json``.
    kwargs: ``gpu`` (device index, esrgan.py:156), ``dataset`` (a torch Dataset yielding {"lr","hr"}; default synthetic)."""
    given = _opt_dict(opt)
    opt = _complete(opt)
    _check_supported(opt)
    if opt.lambda_hist > 0:
        assert opt.warmup_batches > 0, "if distribution learning is enabled, warmup_batches needs to be greater than 0."   # esrgan.py:159-160
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", kwargs.get("gpu", 0)))
    if not torch.cuda.is_available():
        raise RuntimeError("super-resolution_amd.esrgan.train needs a ROCm GPU (no CPU fallback for the hot path)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", device_id=device)
    model_name = "" if not opt.name else opt.name + "_"
    out_dir = os.path.join(opt.root, opt.model_path)
    info_path = os.path.join(out_dir, model_name + "info.json")
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
    info = {"epochs": 0, "argument": given}
    if opt.set_seed > 0:
        torch.manual_seed(opt.set_seed)
        np.random.seed(opt.set_seed)
        info["seed"] = opt.set_seed
    else:
        torch.manual_seed(0)            # replicas must start identical on every rank

    st = _train.Stepper(workload="gan", res_blocks=opt.residual_blocks, device=device, factor=opt.factor,
                        distributed=(world > 1), channels=opt.channels, res_scale=opt.res_scale,
                        lr=opt.lr, lr_g=opt.lr_g, lr_d=opt.lr_d, betas=(opt.b1, opt.b2), weight_decay=opt.l2decay,
                        d_channels=tuple(opt.d_channels), lambdas=(opt.lambda_pix, opt.lambda_pow), lambda_hr=opt.lambda_hr,
                        lambda_adv=opt.lambda_adv, lambda_lr=opt.lambda_lr, lambda_reg=opt.lambda_reg, d_threshold=opt.d_threshold,
                        scaling_power=opt.scaling_power, multiplier=opt.pixel_multiplier, hr_shape=(opt.hr_height, opt.hr_width),
                        num_final_layer_res=opt.num_final_res_blocks, uniform_init=opt.uniform_init, lambda_nnz=opt.lambda_nnz,
                        lambda_mask=opt.lambda_mask, lambda_hit=opt.lambda_hit, lambda_hist=opt.lambda_hist,
                        hit_threshold=opt.hit_threshold, sigma=opt.sigma, conditional=opt.conditional, drop_rate=opt.drop_rate, discriminator=opt.discriminator, relativistic=opt.relativistic,
                        use_transposed_conv=opt.use_transposed_conv, fully_tconv_upsample=opt.fully_transposed_conv)
    if opt.E_thres:
        st.generator.thres = opt.E_thres
    load_chk = bool(opt.load_checkpoint)
    if load_chk:
        st.generator.load_state_dict(torch.load(opt.load_checkpoint, map_location=device))
        gfile = os.path.basename(opt.load_checkpoint)
        for k, D in st.discriminators.items():
            dpath = opt.load_checkpoint.replace(gfile, gfile.replace("generator", ["discriminator", "discriminator_pow"][k]))
            if os.path.exists(dpath):
                D.load_state_dict(torch.load(dpath, map_location=device))
        if model_name == "":
            model_name = gfile.split("generator")[0]
            info_path = os.path.join(out_dir, model_name + "info.json")
        if os.path.exists(info_path):
            with open(info_path) as f:
                info = json.load(f)

    dataset = kwargs.get("dataset")
    if dataset is None:
        if opt.dataset_type in ("jet", "spjet"):   # datasets.py:316-329, from a .npy file of the dataframe's rows
            from . import datasets as _ds
            dataset = _ds.get_dataset(opt.dataset_type, opt.dataset_path, opt.hr_height, opt.hr_width, opt.factor, pre=opt.pre_factor,
                                      threshold=opt.E_thres, N=opt.n_hardest, noise_factor=opt.noise_factor)
        elif opt.dataset_type != "synthetic":
            raise NotImplementedError("the HDF5/text event datasets (datasets.py) are outside this build; pass dataset=..., use "
                                      "--dataset_type synthetic, or jet/spjet with a .npy row file")
    if dataset is None:
        dataset = SyntheticJets(opt.synthetic_batches * opt.batch_size, opt.channels, opt.hr_height, opt.hr_width, opt.factor)
    sampler = None
    if world > 1:
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=True, drop_last=True)
    per_rank = max(opt.batch_size // world, 1)
    loader = torch.utils.data.DataLoader(dataset, batch_size=per_rank, shuffle=(sampler is None), sampler=sampler, drop_last=(world > 1))

    loss_dict = info.get("loss") or {k: [] for k in LOSS_KEYS}
    for k in LOSS_KEYS:                      # info.json written before the optional heads existed
        loss_dict.setdefault(k, [])
    nnz = []                                 # non-zero HR pixel values seen during warm-up (esrgan.py:434-437)
    if opt.lambda_hist > 0 and load_chk and "binedges0" in info:
        for k in range(2):
            if "binedges%i" % k in info:
                st.set_hist_binedges(k, np.array(info["binedges%i" % k]))
    batches_trained = int(info.get("batches_done", 0))
    start_epoch = int(info.get("epochs", 0))
    n_batches = math.inf if opt.n_batches == -1 else opt.n_batches
    total_batches = len(loader) * (opt.n_epochs - start_epoch) if n_batches == math.inf else n_batches + batches_trained
    batches_done = batches_trained - 1
    save_info_file = opt.save_info if opt.save_info is not None else opt.save

    def save_info():
        if save_info_file and rank == 0:
            info["loss"] = loss_dict
            info["batches_done"] = batches_done
            with open(info_path, "w") as f:
                json.dump(info, f)

    def save_weights(epoch):
        if not opt.save or rank != 0:
            return
        suffix = "_continued" if load_chk else ""
        torch.save(st.generator.state_dict(), os.path.join(out_dir, "%sgenerator_%d%s.pth" % (model_name, epoch, suffix)))
        for k, D in st.discriminators.items():
            torch.save(D.state_dict(), os.path.join(out_dir, "%sdiscriminator%s_%d%s.pth" % (model_name, ["", "_pow"][k], epoch, suffix)))
        save_info()

    for epoch in range(start_epoch, opt.n_epochs + start_epoch):
        if sampler is not None:
            sampler.set_epoch(epoch)
        for i, imgs in enumerate(loader):
            batches_done += 1
            if "rows" in imgs:                      # raw sparse event rows: the whole batch is decoded by one kernel launch
                imgs = dataset.decode_batch(imgs["rows"].to(device))
            imgs_lr = imgs["lr"].to(device).float()
            imgs_hr = imgs["hr"].to(device).float()
            in_warm_branch = (not load_chk) or opt.lambda_hist > 0                      # esrgan.py:417
            warm = in_warm_branch and (batches_done - batches_trained < opt.warmup_batches)
            if in_warm_branch and opt.lambda_hist > 0 and nnz is not None and batches_done - batches_trained == opt.warmup_batches:
                _set_binedges(st, opt, nnz, info)                                       # esrgan.py:440-456
                nnz = None
            if warm:
                if opt.lambda_hist > 0:
                    v = imgs_hr.reshape(-1)
                    nnz.extend(list(v[v > 0].cpu().numpy()))            # float32 items, like esrgan.py:436-437
                if opt.learn_warmup:
                    out = st.warmup_step(imgs_lr, imgs_hr)
                    if batches_done % opt.report_freq == 0:
                        v = out["g_loss"].item()
                        loss_dict["g_loss"].append(v); loss_dict["pixel_loss"].append(v)
                        if rank == 0:
                            print("[Batch %d/%d] [Epoch %d/%d] [G pixel: %f]" % (i, total_batches, epoch, opt.n_epochs, v))
                continue
            do_g = (i == opt.warmup_batches) or (batches_done % opt.update_g == 0)
            do_d = (i == opt.warmup_batches) or (batches_done % opt.update_d == 0)
            out = st.gan_step(imgs_lr, imgs_hr, update_g=do_g, update_d=do_d)
            vals = st.loss_scalars(out)                      # one host sync for the whole report
            # esrgan.py:645-648; under data parallelism `nan_probe` is the all-reduced flag: every rank raises together
            if any(v != v for v in (vals["d_loss_def"], vals["d_loss_pow"], vals["g_loss"], vals["nan_probe"])):
                save_info()
                raise ValueError("loss is NaN\n[Batch %d] [D def: %e, pow: %e] [G loss: %f]" %
                                 (i, vals["d_loss_def"], vals["d_loss_pow"], vals["g_loss"]))
            if batches_done % opt.report_freq == 0:
                for k in LOSS_KEYS:
                    loss_dict[k].append(vals[k])
                if rank == 0:
                    print("[Batch %d] [D def: %f, pow: %f] [G loss: %f [def: %f, pow: %f], adv: %f, adv pow: %f, pixel: %f, "
                          "pixel pow: %f, lr pixel: %f, lr pixel pow: %f, hist: %f, hist pow: %f, nnz: %f, nnz pow: %f, mask: %f, "
                          "mask pow: %f, wasser: %f, wasser pow: %f, hit: %f, hit pow: %f, wasserdist: %f, wasserdist pow: %f]"
                          % ((batches_done,) + tuple(vals[k] for k in LOSS_KEYS)))
            # esrgan.py:280-283,780-783: every checkpoint_interval batches, or n_checkpoints times over the run when that is given
            if opt.n_checkpoints != -1:
                every = max(int(total_batches // opt.n_checkpoints), 1) if total_batches != math.inf else 0
            else:
                every = opt.checkpoint_interval
            if every > 0 and (batches_done + 1) % every == 0:
                save_weights(epoch)
            if opt.save_late > 0 and (batches_done - batches_trained + 1) == opt.save_late:      # esrgan.py:800-804
                if opt.save and rank == 0:
                    torch.save(st.generator.state_dict(), os.path.join(out_dir, "%sgenerator_%s_ep%i.pth" % (model_name, "late_save", epoch)))
                    for k, D in st.discriminators.items():
                        torch.save(D.state_dict(), os.path.join(out_dir, "%sdiscriminator%s_%s_ep%i.pth" % (model_name, ["", "_pow"][k], "late_save", epoch)))
            if batches_done + 1 >= total_batches:
                break
        info["epochs"] = epoch + 1
        if batches_done + 1 >= total_batches:
            break
    save_weights(info["epochs"])
    save_info()
    info["loss"] = loss_dict
    info["batches_done"] = batches_done
    return info


if __name__ == "__main__":
    print(json.dumps({k: v for k, v in train(get_parser()).items() if k != "loss"}, default=str))
