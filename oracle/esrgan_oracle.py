"""CPU oracle for the ESRGAN hot path -- TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU fp32 restatement of the reference's arithmetic
for the hot path.  It is *not* part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it, and only as the checker.  The product path
(``super-resolution_amd``) never imports anything from ``oracle/``.

Parity pin: the functions here are checked against the imported reference
(``/root/reference/models.py``) by ``tools/make_golden.py`` in the build
container, and against the committed fixtures ``tests/golden/*.npz`` by
``tests/test_oracle_golden.py`` everywhere else.

Everything operates on a flat ``dict[str, Tensor]`` with the reference's
``state_dict`` key names (OIHW fp32 weights), NCHW fp32 activations.

Reference citations (file:line in /root/reference):
  models.py:14-41   DenseResidualBlock        -> dense_residual_block
  models.py:44-53   ResidualInResidualDenseBlock -> rrdb
  models.py:56-135  GeneratorRRDB             -> generator_forward
  models.py:140-174 Markovian_Discriminator   -> discriminator_forward
  models.py:297-305 SumPool2d                 -> sum_pool
  esrgan.py:416-439 warm-up step              -> warmup_step
  esrgan.py:457-555 G phase                   -> g_phase_loss
  esrgan.py:561-626 D phase                   -> d_phase_loss
  utils.py:259-272  softgreater / get_hitogram / nnz_mask -> same names
  models.py:308-342 DiffableHistogram         -> diffable_histogram
  utils.py:90-113   KLD_hist                  -> kld_hist
  esrgan.py:434-456 bin edges of the energy histogram -> hist_binedges
  esrgan.py:522-547 optional loss heads       -> g_phase_loss(heads=...)
  models.py:189-223 Conditional_Discriminator -> conditional_discriminator_forward
  datasets.py:136-145,170-201,232-249 extract / cutters / SparseJetDataset item -> extract, *_cutter, sparse_jet_item
"""
import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
G_SLOPE = 0.01   # nn.LeakyReLU() default, models.py:21,75,88,98
D_SLOPE = 0.2    # nn.LeakyReLU(0.2), models.py:143,145
INNER_RES_SCALE = 0.2  # DenseResidualBlock default; RRDB does not forward its arg (models.py:49)


# --------------------------------------------------------------------------- utils
def closed_form_fill(sd: Dict[str, Tensor], gain: float = 1.0) -> Dict[str, Tensor]:
    """Deterministic, storage-free weights: for the li-th key (sorted order) element i
    of a conv weight is ``gain/sqrt(fan_in) * sin(0.37*i + li)``; biases
    ``0.05*sin(0.91*i + li)``.  ``power``/``multiplier`` are left untouched."""
    out = {}
    for li, k in enumerate(sorted(sd.keys())):
        t = sd[k]
        if k in ("power", "multiplier"):
            out[k] = t.clone()
            continue
        n = t.numel()
        idx = torch.arange(n, dtype=torch.float64)
        if t.dim() == 4:
            fan_in = t.shape[1] * t.shape[2] * t.shape[3]
            v = gain / math.sqrt(fan_in) * torch.sin(0.37 * idx + li)
        else:
            v = 0.05 * torch.sin(0.91 * idx + li)
        out[k] = v.to(torch.float32).reshape(t.shape)
    return out


def jet_images(n: int, c: int, h: int, w: int, seed: int, factor: int):
    """Synthetic jet-like sparse non-negative HR images and their LR sum-pool
    (LR construction = datasets.py:227,247 SumPool2d)."""
    g = torch.Generator().manual_seed(seed)
    hr = 10.0 * torch.rand(n, c, h, w, generator=g) * (torch.rand(n, c, h, w, generator=g) < 0.1).float()
    lr = sum_pool(hr, factor)
    return lr, hr


def lrelu(x: Tensor, slope: float) -> Tensor:
    return torch.where(x > 0, x, x * slope)


def conv3x3(x: Tensor, w: Tensor, b: Optional[Tensor], stride: int = 1) -> Tensor:
    """nn.Conv2d(kernel_size=3, stride, padding=1) (models.py:19,63,67,87,97,99,142,144,168)."""
    return F.conv2d(x, w, b, stride=stride, padding=1)


def pixel_shuffle(x: Tensor, r: int = 2) -> Tensor:
    """nn.PixelShuffle(r) (models.py:89): out[n,c,r*h+i,r*w+j] = in[n,c*r*r+r*i+j,h,w]."""
    n, c, h, w = x.shape
    co = c // (r * r)
    x = x.reshape(n, co, r, r, h, w)
    x = x.permute(0, 1, 4, 2, 5, 3)
    return x.reshape(n, co, h * r, w * r)


def sum_pool(x: Tensor, k: int) -> Tensor:
    """SumPool2d (models.py:297-305): k*k * AvgPool2d(k)."""
    return (k * k) * F.avg_pool2d(x, k)


# --------------------------------------------------------------------------- generator
DROP_MASKS: Optional[dict] = None    # {dense-block prefix: [N, F, 1, 1] mask already divided by (1 - p)} -> Dropout2d of models.py:38-39


def dense_residual_block(sd, prefix: str, x: Tensor) -> Tensor:
    """models.py:34-41; with DROP_MASKS set, the channel dropout of ``drop_rate > 0`` is applied with the given masks."""
    inputs = x
    out = x
    for k in range(1, 6):
        out = conv3x3(inputs, sd[f"{prefix}.b{k}.0.weight"], sd[f"{prefix}.b{k}.0.bias"])
        if k < 5:
            out = lrelu(out, G_SLOPE)
        inputs = torch.cat([inputs, out], 1)
    if DROP_MASKS is not None and prefix in DROP_MASKS:
        out = out * DROP_MASKS[prefix]
    return out * INNER_RES_SCALE + x


def rrdb(sd, prefix: str, x: Tensor, res_scale: float) -> Tensor:
    """models.py:52-53."""
    out = x
    for j in range(3):
        out = dense_residual_block(sd, f"{prefix}.dense_blocks.{j}", out)
    return out * res_scale + x


def _out(x: Tensor, thres: float, pw: float, training: bool) -> Tensor:
    """GeneratorRRDB.out, models.py:114-118."""
    lambd = float(thres) ** float(pw)
    if training:
        return F.hardshrink(x, lambd=lambd)
    return F.hardshrink(F.relu(x), lambd=lambd)


def upsample_kinds(num_upsample: int, use_transposed_conv: bool = False, fully_tconv_upsample: bool = False):
    """models.py:69-90: per upsampling stage "shuffle" (conv F->4F, LeakyReLU, PixelShuffle(2)) or "tconv"
    (ConvTranspose2d(F, F, 2, stride 2), LeakyReLU): odd stages with use_transposed_conv, every stage with fully_tconv_upsample
    (use_transposed_conv takes precedence: it is tested first, models.py:70)."""
    if use_transposed_conv:
        return ["tconv" if u % 2 == 1 else "shuffle" for u in range(num_upsample)]
    return ["tconv" if fully_tconv_upsample else "shuffle"] * num_upsample


def generator_forward(sd, x: Tensor, num_res_blocks: int, num_upsample: int,
                      res_scale: float = 0.2, training: bool = True, thres: float = 0.0,
                      num_final_layer_res: int = 0, use_transposed_conv: bool = False,
                      fully_tconv_upsample: bool = False) -> Tuple[Tensor, Tensor]:
    """GeneratorRRDB.forward, models.py:120-135 (upsampling branches 69-90; nn.Sequential indices: a "shuffle" stage takes
    three slots, a "tconv" stage two).  Returns (output, srs)."""
    power = sd["power"]
    mult = sd["multiplier"]
    x = mult * (x ** power)
    out1 = conv3x3(x, sd["conv1.weight"], sd["conv1.bias"])
    out = out1
    for i in range(num_res_blocks):
        out = rrdb(sd, f"res_blocks.{i}", out, res_scale)
    out2 = conv3x3(out, sd["conv2.weight"], sd["conv2.bias"])
    out = out1 + out2
    idx = 0
    for kind in upsample_kinds(num_upsample, use_transposed_conv, fully_tconv_upsample):
        if kind == "shuffle":
            out = conv3x3(out, sd[f"upsampling.{idx}.weight"], sd[f"upsampling.{idx}.bias"])
            out = pixel_shuffle(lrelu(out, G_SLOPE), 2)
            idx += 3
        else:
            out = lrelu(F.conv_transpose2d(out, sd[f"upsampling.{idx}.weight"], sd[f"upsampling.{idx}.bias"], stride=2), G_SLOPE)
            idx += 2
    if num_final_layer_res > 0:
        o3 = out
        for i in range(num_final_layer_res):
            o3 = rrdb(sd, f"res_blocks_final.{i}", o3, res_scale)
        out = o3 + out
    out = conv3x3(out, sd["conv3.0.weight"], sd["conv3.0.bias"])
    out = lrelu(out, G_SLOPE)
    out = conv3x3(out, sd["conv3.2.weight"], sd["conv3.2.bias"]) / mult
    srs = _out(out, thres, float(power), training)
    if float(power) != 1:
        out = F.relu(out) ** (1 / power)
    return _out(out, thres, 1.0, training), srs


def generator_state_shapes(channels=1, filters=64, num_res_blocks=10, num_upsample=1,
                           num_final_layer_res=0, use_transposed_conv=False, fully_tconv_upsample=False) -> Dict[str, Tuple[int, ...]]:
    """state_dict key -> shape for GeneratorRRDB (models.py:58-106)."""
    F_ = filters
    sh = {"power": (1,), "multiplier": (1,),
          "conv1.weight": (F_, channels, 3, 3), "conv1.bias": (F_,)}

    def add_rrdbs(name, n):
        for i in range(n):
            for j in range(3):
                for k in range(1, 6):
                    p = f"{name}.{i}.dense_blocks.{j}.b{k}.0"
                    sh[p + ".weight"] = (F_, k * F_, 3, 3)
                    sh[p + ".bias"] = (F_,)
    add_rrdbs("res_blocks", num_res_blocks)
    sh["conv2.weight"] = (F_, F_, 3, 3)
    sh["conv2.bias"] = (F_,)
    idx = 0
    for kind in upsample_kinds(num_upsample, use_transposed_conv, fully_tconv_upsample):
        if kind == "shuffle":
            sh[f"upsampling.{idx}.weight"] = (4 * F_, F_, 3, 3)
            sh[f"upsampling.{idx}.bias"] = (4 * F_,)
            idx += 3
        else:
            sh[f"upsampling.{idx}.weight"] = (F_, F_, 2, 2)       # ConvTranspose2d: (in, out, kH, kW)
            sh[f"upsampling.{idx}.bias"] = (F_,)
            idx += 2
    add_rrdbs("res_blocks_final", num_final_layer_res)
    sh["conv3.0.weight"] = (F_, F_, 3, 3)
    sh["conv3.0.bias"] = (F_,)
    sh["conv3.2.weight"] = (channels, F_, 3, 3)
    sh["conv3.2.bias"] = (channels,)
    return sh


def default_init_generator(seed: int, **cfg) -> Dict[str, Tensor]:
    """PyTorch default Conv2d init (kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in)),
    bias U(+-1/sqrt(fan_in))) in state-dict key order, own RNG stream."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    shapes = generator_state_shapes(**cfg)
    for k, s in shapes.items():
        if k == "power" or k == "multiplier":
            sd[k] = torch.ones(1)
        elif k.endswith("weight"):
            bound = 1.0 / math.sqrt(s[1] * 9)
            sd[k] = (torch.rand(s, generator=g) * 2 - 1) * bound
        else:
            wk = k[:-4] + "weight"
            bound = 1.0 / math.sqrt(shapes[wk][1] * 9)
            sd[k] = (torch.rand(s, generator=g) * 2 - 1) * bound
    return sd


# --------------------------------------------------------------------------- discriminator
def discriminator_state_shapes(in_channels=1, channels=(16, 32, 32, 64)):
    """Markovian_Discriminator (models.py:149-171): model.{0,2,...}."""
    sh = {}
    idx = 0
    cin = in_channels
    for co in channels:
        sh[f"model.{idx}.weight"] = (co, cin, 3, 3); sh[f"model.{idx}.bias"] = (co,)
        sh[f"model.{idx+2}.weight"] = (co, co, 3, 3); sh[f"model.{idx+2}.bias"] = (co,)
        idx += 4
        cin = co
    sh[f"model.{idx}.weight"] = (1, cin, 3, 3); sh[f"model.{idx}.bias"] = (1,)
    return sh


def discriminator_output_shape(input_shape, channels=(16, 32, 32, 64)):
    """models.py:156-170: ceil(H/2) per block."""
    _, h, w = input_shape
    for _ in channels:
        h = int(math.ceil(h / 2)); w = int(math.ceil(w / 2))
    return (1, h, w)


def discriminator_forward(sd, img: Tensor, channels=(16, 32, 32, 64)) -> Tensor:
    """Markovian_Discriminator.forward (models.py:173-174) over discriminator_block (140-146)."""
    x = img
    idx = 0
    for _ in channels:
        x = lrelu(conv3x3(x, sd[f"model.{idx}.weight"], sd[f"model.{idx}.bias"], 1), D_SLOPE)
        x = lrelu(conv3x3(x, sd[f"model.{idx+2}.weight"], sd[f"model.{idx+2}.bias"], 2), D_SLOPE)
        idx += 4
    return conv3x3(x, sd[f"model.{idx}.weight"], sd[f"model.{idx}.bias"], 1)


def standard_discriminator_forward(sd, img: Tensor, channels=(16, 32, 32, 64)) -> Tensor:
    """Standard_Discriminator.forward (models.py:177-186): the patch trunk WITHOUT its final 1-channel conv (``model[:-1]``
    keeps the trailing LeakyReLU), flattened into Linear(., 1024) -> ReLU -> Linear(1024, 1)."""
    x = img
    idx = 0
    for _ in channels:
        x = lrelu(conv3x3(x, sd[f"model.{idx}.weight"], sd[f"model.{idx}.bias"], 1), D_SLOPE)
        x = lrelu(conv3x3(x, sd[f"model.{idx+2}.weight"], sd[f"model.{idx+2}.bias"], 2), D_SLOPE)
        idx += 4
    x = x.reshape(img.shape[0], -1)
    x = F.relu(F.linear(x, sd["fc.0.weight"], sd["fc.0.bias"]))
    return F.linear(x, sd["fc.2.weight"], sd["fc.2.bias"])


def conditional_discriminator_forward(sd, img: Tensor, cond: Tensor, channels=(32, 64, 128, 256), num_upsample=3) -> Tensor:
    """Conditional_Discriminator.forward (models.py:189-223): the HR image goes through ``num_upsample`` stride-(1,2)
    blocks (model_hr), the LR condition through as many stride-(1,1) blocks (model_c); their outputs are concatenated
    on the channel axis and finished by the remaining blocks + the 1-channel conv (endmodel)."""
    def block(x, prefix, idx, s2):
        x = lrelu(conv3x3(x, sd[f"{prefix}.{idx}.weight"], sd[f"{prefix}.{idx}.bias"], 1), D_SLOPE)
        return lrelu(conv3x3(x, sd[f"{prefix}.{idx+2}.weight"], sd[f"{prefix}.{idx+2}.bias"], s2), D_SLOPE)
    h, c = img, cond
    for i in range(min(num_upsample, len(channels))):
        h = block(h, "model_hr", 4 * i, 2)
        c = block(c, "model_c", 4 * i, 1)
    x = torch.cat([h, c], 1)
    idx = 0
    for i in range(num_upsample, len(channels)):
        x = block(x, "endmodel", idx, 2)
        idx += 4
    return conv3x3(x, sd[f"endmodel.{idx}.weight"], sd[f"endmodel.{idx}.bias"], 1)


# --------------------------------------------------------------------------- jet data decode
def extract(data: Tensor, etaBins: int, phiBins: int, channels: int = 1) -> Tensor:
    """datasets.py:136-145: data [2, L] (positions, energies) -> [channels, etaBins, phiBins]; sequential accumulate, stop
    after the first zero energy."""
    rec = torch.zeros((channels, etaBins, phiBins))
    for i in range(data.shape[1]):
        pos = data[0, i]
        phi = int(pos // etaBins)
        eta = int(pos % etaBins)
        rec[0, eta, phi] += data[1, i]
        if data[1, i] == 0:
            break
    return rec.float()


def threshold_cutter(x: Tensor, thres) -> Tensor:
    """datasets.py:170-175."""
    return torch.where(x > thres, x, torch.zeros_like(x))


def n_hardest_cutter(x: Tensor, n: int) -> Tensor:
    """datasets.py:178-186."""
    highest = torch.sort(x.view(-1))[0][-n]
    return torch.where(x >= highest, x, torch.zeros_like(x))


def sparse_jet_item(row: Tensor, etaBins, phiBins, factor, pre_factor=1, threshold=None, n_hardest=None):
    """SparseJetDataset.__getitem__ (datasets.py:236-249, noise_factor=None): dataframe row -> (lr, hr)."""
    img = extract(row[:-1].view(-1, 2).t(), etaBins * pre_factor, phiBins * pre_factor)
    if threshold:
        img = threshold_cutter(img, threshold)
    elif n_hardest:
        img = n_hardest_cutter(img, n_hardest)
    img = img[None, ...]
    if pre_factor > 1:
        img = sum_pool(img, pre_factor)
    return sum_pool(img, factor)[0], img[0].clone()


# --------------------------------------------------------------------------- train-step losses
EPS = 1e-7  # esrgan.py:319


def bce_logits(x: Tensor, target: Tensor) -> Tensor:
    return F.binary_cross_entropy_with_logits(x, target)


def warmup_loss(gen_hr: Tensor, hr: Tensor) -> Tensor:
    """esrgan.py:424: L1(G(lr), hr)."""
    return (gen_hr - hr).abs().mean()


# --------------------------------------------------------------------------- optional physics loss heads
def softgreater(x: Tensor, val, sigma=5000, delta=0) -> Tensor:
    """utils.py:259-261."""
    return torch.sigmoid(sigma * (x - val + delta))


def nnz_mask(x: Tensor, sigma=5e4) -> Tensor:
    """utils.py:271-272."""
    return torch.sigmoid(sigma * x)


def get_hitogram(t: Tensor, factor: int, threshold=.1, sig=80) -> Tensor:
    """utils.py:264-268: cut [B,C,H,W] into factor x factor super-pixels, stack them on the batch axis, average the
    (soft) hit indicator over batch and channel -> [factor, factor]."""
    blocks = torch.cat(torch.split(torch.cat(torch.split(t, factor, -2)), factor, -1))
    if sig > 0:
        return torch.sigmoid(sig * (blocks - threshold)).mean((0, 1))
    return blocks.mean((0, 1))


def diffable_histogram(x: Tensor, binedges, sigma) -> Tensor:
    """models.py:308-342 with a sequence of bin edges (the form esrgan.py:454 builds), batchwise=False: [*] -> [1, K]."""
    edges = torch.as_tensor(binedges, dtype=torch.float64)
    delta = (edges[1:] - edges[:-1]).float()[None, :]
    centers = edges[:-1].float() + .5 * delta
    v = x.reshape(1, -1)
    v = v[:, None, :] - centers[..., None]
    v = torch.sigmoid(sigma * (v + delta[..., None] / 2)) - torch.sigmoid(sigma * (v - delta[..., None] / 2))
    return v.sum(2)


def kld_hist(q_entries: Tensor, p_entries: Tensor, binedges) -> Tensor:
    """utils.py:90-113."""
    edges = torch.as_tensor(binedges)
    binsizes = (edges[1:] - edges[:-1]).float()
    n_p, n_q = p_entries.sum().float(), q_entries.sum().float()
    p = p_entries * binsizes / n_p
    q = ((q_entries + 1e-6) * binsizes / n_q).log()
    return F.kl_div(q, p, reduction='sum') / binsizes.mean()


def hist_binedges(nnz, bins: int, power: float = 1.0):
    """esrgan.py:441-452: bin edges from the non-zero HR pixel values collected during warm-up (90 % quantile cut,
    k-means cluster centres as bin centres).  ``nnz``: 1-D numpy array.  Note esrgan.py:448 sorts the UNPOWERED values."""
    import numpy as np
    from sklearn.cluster import KMeans
    nnz = np.asarray(nnz)
    c, b = np.histogram(nnz ** power, 100)
    e_max = b[(np.cumsum(c) > len(nnz ** power) * .9).argmax()]
    sorted_nnz = np.sort(nnz)
    sorted_nnz = sorted_nnz[sorted_nnz <= e_max]
    k_mean = KMeans(n_clusters=bins, random_state=0).fit(sorted_nnz.reshape(-1, 1))
    centers = np.sort(k_mean.cluster_centers_.flatten())
    return np.array([0, *(np.diff(centers) / 2 + centers[:-1]), e_max])


def g_phase_loss(generated: Sequence[Tensor], hr: Tensor, lr: Tensor, d_sds: Sequence[dict],
                 factor: int, scaling_power: float = 1.0, lambdas=(0.2, 1.0),
                 lambda_hr=1.0, lambda_adv=0.01, lambda_lr=0.1,
                 d_channels=(16, 32, 32, 64), heads: Optional[dict] = None, cond_num_upsample: Optional[int] = None,
                 relativistic: bool = True):
    """esrgan.py:468-552 (relativistic; hr+lr+adv terms, plus the optional heads of esrgan.py:522-547 when ``heads`` =
    dict(lambda_nnz, lambda_mask, lambda_hit, hit_threshold, sigma, lambda_hist, binedges=[edges_def, edges_pow]) is given).
    ``generated`` = [G(lr), G.srs].  Returns (loss_G, dict of parts)."""
    ground_truth = [hr, hr ** scaling_power]
    gen_lr = sum_pool(generated[0], factor)
    generated_lr = [gen_lr, gen_lr ** scaling_power]
    ground_truth_lr = [lr, lr ** scaling_power]
    loss_G = torch.zeros(1)
    parts = {}
    for k in range(2):
        if lambdas[k] <= 0:
            continue
        loss_pixel = (generated[k].mean(0)[None] - ground_truth[k].mean(0)[None]).abs().mean()
        loss_lr = (generated_lr[k] - ground_truth_lr[k]).abs().mean()
        if cond_num_upsample is not None:      # Conditional_Discriminator(img, lr) (esrgan.py:493-494)
            pred_real = conditional_discriminator_forward(d_sds[k], ground_truth[k], ground_truth_lr[k], d_channels, cond_num_upsample).detach()
            pred_fake = conditional_discriminator_forward(d_sds[k], generated[k], generated_lr[k], d_channels, cond_num_upsample)
        else:
            pred_real = discriminator_forward(d_sds[k], ground_truth[k], d_channels).detach()
            pred_fake = discriminator_forward(d_sds[k], generated[k], d_channels)
        valid = torch.ones_like(pred_real)
        fake = torch.zeros_like(pred_real)
        if relativistic:
            loss_gan = 0.5 * (bce_logits(EPS + pred_fake - pred_real.mean(0, keepdim=True), valid) +
                              bce_logits(EPS + pred_real - pred_fake.mean(0, keepdim=True), fake))
        else:                                               # esrgan.py:509-510
            loss_gan = bce_logits(EPS + pred_fake, valid)
        tot = lambda_hr * loss_pixel + lambda_adv * loss_gan + lambda_lr * loss_lr
        parts[k] = dict(pixel=loss_pixel, lr=loss_lr, adv=loss_gan)
        h = heads or {}
        if h.get("lambda_nnz", 0) > 0:                                             # esrgan.py:522-525
            gen_nnz = softgreater(generated[k], 0, 50000).sum(1).sum(1).sum(1)
            target = (ground_truth[k] > 0).sum(1).sum(1).sum(1).float()
            parts[k]["nnz"] = F.mse_loss(gen_nnz, target)
            tot = tot + h["lambda_nnz"] * parts[k]["nnz"]
        if h.get("lambda_mask", 0) > 0:                                            # esrgan.py:526-529
            parts[k]["mask"] = (nnz_mask(generated[k]) - nnz_mask(ground_truth[k])).abs().mean()
            tot = tot + h["lambda_mask"] * parts[k]["mask"]
        if h.get("lambda_hist", 0) > 0:                                            # esrgan.py:530-538
            edges = h["binedges"][k]
            gen_hist = diffable_histogram(generated[k][generated[k] > 0], edges, h["sigma"])
            real_hist = diffable_histogram(ground_truth[k][ground_truth[k] > 0], edges, h["sigma"])
            parts[k]["hist"] = kld_hist(gen_hist, real_hist, edges)
            tot = tot + h["lambda_hist"] * parts[k]["hist"]
        if h.get("lambda_hit", 0) > 0:                                             # esrgan.py:543-547
            gen_hit = get_hitogram(generated[k], factor, h["hit_threshold"], h["sigma"])
            target = get_hitogram(ground_truth[k], factor, h["hit_threshold"], h["sigma"])
            parts[k]["hit"] = F.mse_loss(gen_hit, target)
            tot = tot + h["lambda_hit"] * parts[k]["hit"]
        loss_G = loss_G + lambdas[k] * tot
        parts[k]["tot"] = tot
    return loss_G, parts


def d_phase_loss(d_sd: dict, gt: Tensor, gen_detached: Tensor, epsilon: Optional[Tensor],
                 lambda_reg=0.01, d_channels=(16, 32, 32, 64), cond: Optional[Tensor] = None, num_upsample: int = 0,
                 relativistic: bool = True):
    """esrgan.py:569-606 for one discriminator (relativistic + gradient penalty).
    ``epsilon``: (B,1,1,1) interpolation factors (esrgan.py:598) or None to skip GP.
    ``cond`` (the LR ground truth, esrgan.py:569-570,601) selects the Conditional_Discriminator."""
    if cond is not None:
        def discriminator_forward(sd, x, ch):          # noqa: F811  (conditional variant, same call sites)
            return conditional_discriminator_forward(sd, x, cond, ch, num_upsample)
    else:
        discriminator_forward = globals()["discriminator_forward"]
    pred_real = discriminator_forward(d_sd, gt, d_channels)
    pred_fake = discriminator_forward(d_sd, gen_detached, d_channels)
    valid = torch.ones_like(pred_real)
    fake = torch.zeros_like(pred_real)
    if relativistic:
        loss_real = bce_logits(EPS + pred_real - pred_fake.mean(0, keepdim=True), valid)
        loss_fake = bce_logits(EPS + pred_fake - pred_real.mean(0, keepdim=True), fake)
    else:                                                   # esrgan.py:584-586
        loss_real = bce_logits(EPS + pred_real, valid)
        loss_fake = bce_logits(EPS + pred_fake, fake)
    loss_D = (loss_real + loss_fake) / 2
    gp = None
    if lambda_reg > 0 and epsilon is not None:
        interp = (epsilon * gt + (1 - epsilon) * gen_detached).detach().requires_grad_(True)
        pred_i = discriminator_forward(d_sd, interp, d_channels)
        grads = torch.autograd.grad(pred_i, interp, grad_outputs=valid, create_graph=True,
                                    retain_graph=True, only_inputs=True)[0]
        grads = grads.view(gt.shape[0], -1)
        gp = ((grads.norm(2, dim=1) - 1) ** 2).mean() * lambda_reg / 2
        loss_D = loss_D + gp
    return loss_D, gp


# --------------------------------------------------------------------------- eval-mode metrics
def calculate_metrics(sd, batches, num_res_blocks, num_upsample, res_scale, factor):
    """HR/LR L1 part of evaluation/eval.py:455-494 (no EMD): eval-mode forward, SumPool, per-image L1 means, and the
    reference's key assignment (eval.py:491 zips ['hr_l1','lr_l1'] with [lr_similarity, hr_similarity])."""
    import numpy as np
    lr_sim, hr_sim = [], []
    with torch.no_grad():
        for lr, hr in batches:
            gen_hr, _ = generator_forward(sd, lr, num_res_blocks, num_upsample, res_scale, training=False)
            gen_lr = sum_pool(gen_hr, factor)
            lr_sim.extend((gen_lr - lr).abs().numpy().mean((1, 2, 3)).tolist())
            hr_sim.extend((gen_hr - hr).abs().numpy().mean((1, 2, 3)).tolist())
    return {n: {"mean": float(np.mean(v)), "std": float(np.std(v))} for n, v in zip(["hr_l1", "lr_l1"], [lr_sim, hr_sim])}
