"""ImageNet loaders with the interface of the reference's ``utils/datasets.py`` (LoaderGenerator :24-63,
ImageNetLoaderGenerator :66-97, CacheDataset :99-108, ViTImageNetLoaderGenerator :111-125) -- without torchvision / timm, which
this environment does not have: folders are walked with ``os``, images decoded with PIL, transforms are numpy / torch.

What the reference takes from timm (``resolve_data_config(model.default_cfg)`` + ``create_transform``) is restated here as a
per-family table of the timm 0.9.2 defaults (interpolation, crop fraction, mean / std).  Follow-the-source restatement:
timm is not importable here, so these numbers are **unpinned** against it; the evaluation transform is
the standard resize(size / crop_pct) -> centre crop -> normalise (resize arithmetic as torchvision's: long side truncated), the
training-side transform used for the calibration subset is random-resized-crop + horizontal flip.  KNOWN DEVIATION: timm's
training transform also applies colour jitter (0.4), which is left out here -- calibration images, and with them Prec@1, do not
reproduce the reference bit for bit (test_quant.py logs this once).
"""
import math
import os

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
IMAGENET_INCEPTION_MEAN, IMAGENET_INCEPTION_STD = (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)
_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")


def data_config(model_name: str) -> dict:
    """Stand-in for timm's resolve_data_config for the reference's model zoo."""
    size = 384 if model_name.endswith("384") else 224
    if model_name.startswith("vit_"):                      # augreg ViTs: inception statistics, crop 0.9
        return dict(input_size=size, interpolation="bicubic", crop_pct=0.9, mean=IMAGENET_INCEPTION_MEAN, std=IMAGENET_INCEPTION_STD)
    if model_name.startswith("deit_"):
        return dict(input_size=size, interpolation="bicubic", crop_pct=0.875, mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD)
    if model_name.startswith("swin_"):
        return dict(input_size=size, interpolation="bicubic", crop_pct=1.0 if size == 384 else 0.9,
                    mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD)
    raise ValueError(f"no data config for {model_name}")


def _resample(name):
    from PIL import Image
    return {"bicubic": Image.BICUBIC, "bilinear": Image.BILINEAR, "nearest": Image.NEAREST}[name]


def _to_tensor(img, mean, std):
    a = np.asarray(img, dtype=np.float32) / 255.0                       # HWC in [0, 1]
    t = torch.from_numpy(a).permute(2, 0, 1).contiguous()
    return (t - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)


def _resize_short(img, size, resample):
    """torchvision.transforms.Resize(int): the short side becomes `size`, the long side int(size * long / short) -- TRUNCATED,
    as torchvision computes it (functional._compute_resized_output_size)."""
    w, h = img.size
    if (w <= h and w == size) or (h <= w and h == size):
        return img
    if w < h:
        return img.resize((size, int(size * h / w)), resample)
    return img.resize((int(size * w / h), size), resample)


class EvalTransform:
    """resize the short side to round(input_size / crop_pct), centre crop, to tensor, normalise"""

    def __init__(self, input_size=224, interpolation="bicubic", crop_pct=0.875, mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD):
        self.size, self.resample, self.mean, self.std = input_size, _resample(interpolation), mean, std
        self.scale_size = int(math.floor(input_size / crop_pct))

    def __call__(self, img):
        img = _resize_short(img, self.scale_size, self.resample)
        w, h = img.size
        left, top = (w - self.size) // 2, (h - self.size) // 2
        return _to_tensor(img.crop((left, top, left + self.size, top + self.size)), self.mean, self.std)


class TrainTransform:
    """random-resized crop (scale 0.08..1, ratio 3/4..4/3) + horizontal flip, to tensor, normalise; draws from torch's RNG"""

    def __init__(self, input_size=224, interpolation="bicubic", crop_pct=0.875, mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD,
                 pre_resize=None):
        self.size, self.resample, self.mean, self.std = input_size, _resample(interpolation), mean, std
        self.pre_resize = pre_resize                      # Resize(256) in front of the crop (datasets.py:71-72, torchvision recipe)

    def __call__(self, img):
        from PIL import Image
        if self.pre_resize:
            img = _resize_short(img, self.pre_resize, self.resample)
        w, h = img.size
        area = w * h
        box = None
        for _ in range(10):
            target = area * float(torch.empty(1).uniform_(0.08, 1.0))
            logr = torch.empty(1).uniform_(math.log(3 / 4), math.log(4 / 3))
            ratio = math.exp(float(logr))
            cw, ch = int(round(math.sqrt(target * ratio))), int(round(math.sqrt(target / ratio)))
            if 0 < cw <= w and 0 < ch <= h:
                top, left = int(torch.randint(0, h - ch + 1, (1,))), int(torch.randint(0, w - cw + 1, (1,)))
                box = (left, top, left + cw, top + ch)
                break
        if box is None:                                                   # fallback: central crop of the whole image
            s = min(w, h)
            box = ((w - s) // 2, (h - s) // 2, (w - s) // 2 + s, (h - s) // 2 + s)
        img = img.resize((self.size, self.size), self.resample, box=box)
        if float(torch.rand(1)) < 0.5:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        return _to_tensor(img, self.mean, self.std)


class ImageFolder(Dataset):
    """``root/<class>/<image>``: classes are the sorted sub-directory names (torchvision.datasets.ImageFolder's rule)"""

    def __init__(self, root, transform=None):
        self.root, self.transform = root, transform
        self.classes = sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
        if not self.classes:
            raise FileNotFoundError(f"no class directories under {root}")
        self.class_to_idx = {c: i for i, c in enumerate(self.classes)}
        self.samples = []
        for c in self.classes:
            for dirpath, _, files in sorted(os.walk(os.path.join(root, c))):
                for f in sorted(files):
                    if f.lower().endswith(_EXTENSIONS):
                        self.samples.append((os.path.join(dirpath, f), self.class_to_idx[c]))
        if not self.samples:
            raise FileNotFoundError(f"no images under {root}")

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, i):
        from PIL import Image
        path, target = self.samples[i]
        with open(path, "rb") as f:
            img = Image.open(f).convert("RGB")
        return (self.transform(img) if self.transform else img), target


class CacheDataset(Dataset):
    def __init__(self, datas) -> None:
        super().__init__()
        self.datas = datas

    def __getitem__(self, idx):
        return self.datas[idx]

    def __len__(self):
        return len(self.datas)


class LoaderGenerator:
    """datasets.py:24-63: ``val_loader()`` and ``calib_loader(num, batch_size, seed)`` (a seeded random subset of the
    training set, pre-loaded into memory)."""

    def __init__(self, root, val_batch_size=1, num_workers=0, kwargs={}):
        self.root, self.val_batch_size, self.num_workers, self.kwargs = root, val_batch_size, num_workers, dict(kwargs)
        self._train_set = self._val_set = self._calib_set = None
        self.train_transform = self.val_transform = None
        self.train_loader_kwargs = {'num_workers': self.num_workers, 'pin_memory': True, 'drop_last': False}
        self.val_loader_kwargs = {'num_workers': self.num_workers, 'pin_memory': False, 'drop_last': False}
        self.load()

    def load(self):
        pass

    @property
    def train_set(self):
        if self._train_set is None:
            self._train_set = ImageFolder(os.path.join(self.root, 'train'), self.train_transform)
        return self._train_set

    @property
    def val_set(self):
        if self._val_set is None:
            self._val_set = ImageFolder(os.path.join(self.root, 'val'), self.val_transform)
        return self._val_set

    def val_loader(self):
        return DataLoader(self.val_set, batch_size=self.val_batch_size, shuffle=False, **self.val_loader_kwargs)

    def calib_loader(self, num=1024, batch_size=32, seed=3, in_memory=True):
        np.random.seed(seed)
        inds = np.random.permutation(len(self.train_set))[:num]
        if in_memory:
            self._calib_set = CacheDataset([self.train_set[int(i)] for i in inds])
        else:
            import copy
            sub = copy.copy(self.train_set)
            sub.transform = self.val_transform
            self._calib_set = torch.utils.data.Subset(sub, [int(i) for i in inds])
        return DataLoader(self._calib_set, batch_size=batch_size, shuffle=False, **self.train_loader_kwargs)


class ImageNetLoaderGenerator(LoaderGenerator):
    """datasets.py:66-97: the torchvision recipe (resize 256, crop 224, ImageNet statistics; bilinear)"""

    def load(self):
        cfg = dict(input_size=224, interpolation="bilinear", crop_pct=0.875, mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD)
        self.train_transform, self.val_transform = TrainTransform(**cfg, pre_resize=256), EvalTransform(**cfg)


class ViTImageNetLoaderGenerator(ImageNetLoaderGenerator):
    """datasets.py:111-125: the transform that belongs to the model (``kwargs['model']``: a zoo name, or a model object
    carrying ``zoo_name``)."""

    def __init__(self, root, val_batch_size, num_workers, kwargs={}):
        super().__init__(root, val_batch_size=val_batch_size, num_workers=num_workers, kwargs=kwargs)

    def load(self):
        model = self.kwargs.get("model", None)
        assert model is not None, "No model in ViTImageNetLoaderGenerator!"
        name = model if isinstance(model, str) else getattr(model, "zoo_name", None)
        assert name is not None, "ViTImageNetLoaderGenerator: pass the zoo name of the model"
        cfg = data_config(name)
        self.train_transform, self.val_transform = TrainTransform(**cfg), EvalTransform(**cfg)
