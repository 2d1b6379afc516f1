"""Thin harness with the shape of the reference's callers of the path (NOT part of the accelerated path):
    system.py:172-306   NeRFSystem (forward in ray chunks, training_step, validation_step)
    losses.py:10-20     MSELoss(coarse) + MSELoss(fine);  metrics.py:4-13 mse / psnr
    utils/__init__.py:11-50, :56-86   get_optimizer / get_scheduler / extract_model_state_dict / load_ckpt
    eval.py:70-103      batched_inference
pytorch-lightning is not installed in this image; NeRFSystem derives from LightningModule when it is importable
and from torch.nn.Module otherwise -- the method names and signatures are Lightning's, so the reference's
train.py drives it unchanged where Lightning exists.
"""
from __future__ import annotations

from collections import defaultdict

import torch
from torch import nn

from .nerf import Embedding, NeRF
from .rendering import render_rays

try:                                                   # pragma: no cover - not available offline
    from pytorch_lightning import LightningModule as _Base
except Exception:                                      # noqa: BLE001
    _Base = nn.Module


# ----------------------------------------------------------------------------- losses.py / metrics.py
class MSELoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.loss = nn.MSELoss(reduction='mean')

    def forward(self, inputs, targets):
        loss = self.loss(inputs['rgb_coarse'], targets)
        if 'rgb_fine' in inputs:
            loss = loss + self.loss(inputs['rgb_fine'], targets)
        return loss


from .training import FusedAdam, FusedMSELoss  # noqa: E402

loss_dict = {'mse': MSELoss, 'mse_fused': FusedMSELoss}


def mse(image_pred, image_gt, valid_mask=None, reduction='mean'):
    value = (image_pred - image_gt) ** 2
    if valid_mask is not None:
        value = value[valid_mask]
    if reduction == 'mean':
        return torch.mean(value)
    return value


def psnr(image_pred, image_gt, valid_mask=None, reduction='mean'):
    return -10 * torch.log10(mse(image_pred, image_gt, valid_mask, reduction))


# ----------------------------------------------------------------------------- utils/__init__.py
def get_optimizer(hparams, models):
    eps = 1e-8
    parameters = []
    for model in models:
        parameters += list(model.parameters())
    if hparams.optimizer == 'sgd':
        return torch.optim.SGD(parameters, lr=hparams.lr, momentum=hparams.momentum, weight_decay=hparams.weight_decay)
    if hparams.optimizer == 'adam':
        return torch.optim.Adam(parameters, lr=hparams.lr, eps=eps, weight_decay=hparams.weight_decay)
    if hparams.optimizer == 'adam_fused':      # same update as 'adam', one launch per model (training.FusedAdam)
        return FusedAdam(models, lr=hparams.lr, eps=eps, weight_decay=hparams.weight_decay)
    raise ValueError('optimizer not recognized! (sgd / adam; the reference\'s radam / ranger copies are host-side '
                     'code outside the path)')


def get_scheduler(hparams, optimizer):
    if hparams.lr_scheduler == 'steplr':
        return torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=hparams.decay_step, gamma=hparams.decay_gamma)
    if hparams.lr_scheduler == 'cosine':
        return torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=hparams.num_epochs, eta_min=1e-8)
    raise ValueError('scheduler not recognized!')


def extract_model_state_dict(ckpt_path, model_name='model', prefixes_to_ignore=[]):
    """utils/__init__.py:56-71: strip the `<model_name>.` prefix of a (Lightning) checkpoint."""
    checkpoint = torch.load(ckpt_path, map_location=torch.device('cpu'))
    checkpoint_ = {}
    if 'state_dict' in checkpoint:
        checkpoint = checkpoint['state_dict']
    for k, v in checkpoint.items():
        if not k.startswith(model_name):
            continue
        k = k[len(model_name) + 1:]
        if any(k.startswith(p) for p in prefixes_to_ignore):
            continue
        checkpoint_[k] = v
    return checkpoint_


def load_ckpt(model, ckpt_path, model_name='model', prefixes_to_ignore=[]):
    """utils/__init__.py:73-86 (raises instead of exit() on a key mismatch)."""
    model_dict = model.state_dict()
    model_dict.update(extract_model_state_dict(ckpt_path, model_name, prefixes_to_ignore))
    model.load_state_dict(model_dict)


# ----------------------------------------------------------------------------- eval.py:70-103
@torch.no_grad()
def batched_inference(models, embeddings, rays, N_samples, N_importance, use_disp, chunk, white_back):
    """Render a whole image in ray chunks, test_time=True (eval.py:70-103)."""
    B = rays.shape[0]
    results = defaultdict(list)
    for i in range(0, B, chunk):
        rendered = render_rays(models, embeddings, rays[i:i + chunk], N_samples, use_disp, 0, 0, N_importance, chunk,
                               white_back, test_time=True)
        for k, v in rendered.items():
            results[k] += [v]
    return {k: torch.cat(v, 0) for k, v in results.items()}


# ----------------------------------------------------------------------------- system.py:172-306
class NeRFSystem(_Base):
    """hparams needs: loss_type, N_samples, N_importance, use_disp, perturb, noise_std, chunk, optimizer, lr,
    momentum, weight_decay, lr_scheduler, decay_step, decay_gamma, num_epochs (opt.py:3-87).
    Datasets are out of scope (datasets/ needs kornia/cv2): pass `white_back` (blender: True, llff: False) and feed
    {'rays': (B,8), 'rgbs': (B,3)} batches."""

    def __init__(self, hparams, white_back=True):
        super().__init__()
        self.hp = hparams
        self.white_back = white_back
        self.loss = loss_dict[hparams.loss_type]()
        self.embedding_xyz = Embedding(3, 10)
        self.embedding_dir = Embedding(3, 4)
        self.embeddings = [self.embedding_xyz, self.embedding_dir]
        self.nerf_coarse = NeRF()
        self.models = [self.nerf_coarse]
        if hparams.N_importance > 0:
            self.nerf_fine = NeRF()
            self.models += [self.nerf_fine]
        if getattr(hparams, 'pretrained', None):
            load_ckpt(self.nerf_coarse, hparams.pretrained, model_name='nerf_coarse')
            if hparams.N_importance > 0:
                load_ckpt(self.nerf_fine, hparams.pretrained, model_name='nerf_fine')

    def decode_batch(self, batch):
        return batch['rays'], batch['rgbs']

    def forward(self, rays):
        """Batched inference on rays using chunk (system.py:199-223)."""
        rays = rays.reshape(-1, 8)
        B = rays.shape[0]
        results = defaultdict(list)
        for i in range(0, B, self.hp.chunk):
            rendered = render_rays(self.models, self.embeddings, rays[i:i + self.hp.chunk], self.hp.N_samples,
                                   self.hp.use_disp, self.hp.perturb, self.hp.noise_std, self.hp.N_importance,
                                   self.hp.chunk, self.white_back)
            for k, v in rendered.items():
                results[k] += [v]
        return {k: torch.cat(v, 0) for k, v in results.items()}

    def configure_optimizers(self):
        self.optimizer = get_optimizer(self.hp, self.models)
        scheduler = get_scheduler(self.hp, self.optimizer)
        return [self.optimizer], [scheduler]

    def training_step(self, batch, batch_nb):
        rays, rgbs = self.decode_batch(batch)
        results = self(rays)
        loss = self.loss(results, rgbs)
        typ = 'fine' if 'rgb_fine' in results else 'coarse'
        with torch.no_grad():
            psnr_ = psnr(results[f'rgb_{typ}'], rgbs)
        return {'loss': loss, 'progress_bar': {'train_psnr': psnr_}, 'log': {'train/loss': loss, 'train/psnr': psnr_}}

    def validation_step(self, batch, batch_nb):
        rays, rgbs = self.decode_batch(batch)
        rays, rgbs = rays.squeeze(), rgbs.squeeze()
        with torch.no_grad():
            results = self(rays)
        typ = 'fine' if 'rgb_fine' in results else 'coarse'
        return {'val_loss': self.loss(results, rgbs), 'val_psnr': psnr(results[f'rgb_{typ}'], rgbs)}

    def validation_epoch_end(self, outputs):
        mean_loss = torch.stack([x['val_loss'] for x in outputs]).mean()
        mean_psnr = torch.stack([x['val_psnr'] for x in outputs]).mean()
        return {'progress_bar': {'val_loss': mean_loss, 'val_psnr': mean_psnr},
                'log': {'val/loss': mean_loss, 'val/psnr': mean_psnr}}
