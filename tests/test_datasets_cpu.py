"""ImageNet loader / validate stand-ins (reference utils/datasets.py, utils/test_utils.py): folder walk, transforms, the
seeded calibration subset, top-k accuracy -- on a generated toy folder (no torchvision / timm in this environment)."""
import os

import numpy as np
import pytest
import torch

from adalog_amd.utils import datasets as D
from adalog_amd.utils import test_utils as TU


@pytest.fixture()
def toy_imagenet(tmp_path):
    from PIL import Image
    rng = np.random.RandomState(0)
    for split, per in (("train", 5), ("val", 2)):
        for ci, cname in enumerate(("n01", "n02", "n03")):
            d = tmp_path / split / cname
            d.mkdir(parents=True)
            for j in range(per):
                h, w = 60 + 10 * j, 90 - 7 * ci
                arr = (rng.rand(h, w, 3) * 255).astype(np.uint8)
                arr[..., ci] = 255                                   # the class is the saturated channel
                Image.fromarray(arr).save(d / f"img{j}.png")
    return str(tmp_path)


def test_folder_classes_and_eval_transform(toy_imagenet):
    gen = D.ViTImageNetLoaderGenerator(toy_imagenet, val_batch_size=4, num_workers=0, kwargs={"model": "deit_small"})
    vs = gen.val_set
    assert vs.classes == ["n01", "n02", "n03"] and len(vs) == 6
    x, y = vs[0]
    assert x.shape == (3, 224, 224) and x.dtype == torch.float32 and y == 0
    # channel 0 is saturated for class n01: (1 - mean) / std after normalisation
    assert abs(x[0].mean().item() - (1 - 0.485) / 0.229) < 1e-4
    batches = list(gen.val_loader())
    assert [b[0].shape[0] for b in batches] == [4, 2] and batches[0][1].tolist() == [0, 0, 1, 1]


def test_data_config_per_family():
    assert D.data_config("vit_base")["mean"] == (0.5, 0.5, 0.5) and D.data_config("vit_base")["crop_pct"] == 0.9
    assert D.data_config("deit_tiny")["crop_pct"] == 0.875 and D.data_config("swin_base_384")["input_size"] == 384
    t = D.EvalTransform(**D.data_config("swin_small"))
    assert t.scale_size == 248


def test_calib_loader_is_a_seeded_subset(toy_imagenet):
    gen = D.ImageNetLoaderGenerator(toy_imagenet, val_batch_size=2, num_workers=0)
    torch.manual_seed(1)
    a = [y for _, ys in gen.calib_loader(num=6, batch_size=4, seed=3) for y in ys.tolist()]
    torch.manual_seed(1)
    b = [y for _, ys in gen.calib_loader(num=6, batch_size=4, seed=3) for y in ys.tolist()]
    np.random.seed(3)
    want = [gen.train_set.samples[int(i)][1] for i in np.random.permutation(15)[:6]]
    assert a == b == want
    xs = next(iter(gen.calib_loader(num=4, batch_size=4, seed=5)))[0]
    assert xs.shape == (4, 3, 224, 224)


def test_validate_and_accuracy(toy_imagenet):
    gen = D.ImageNetLoaderGenerator(toy_imagenet, val_batch_size=3, num_workers=0)

    class ChannelClassifier(torch.nn.Module):                        # logit c = mean of channel c: right on every toy image
        def forward(self, x):
            return torch.cat([x.mean(dim=(1, 2, 3)).unsqueeze(1) * 0 + x[:, c].mean(dim=(1, 2)).unsqueeze(1) for c in range(3)], 1)

    loss, top1, top5 = TU.validate(gen.val_loader(), ChannelClassifier(), torch.nn.CrossEntropyLoss(), device="cpu")
    assert top1 == 100.0 and top5 == 100.0 and loss > 0
    out = torch.tensor([[0.1, 0.9, 0.0], [0.8, 0.1, 0.1], [0.2, 0.3, 0.5]])
    p1, p2 = TU.accuracy(out, torch.tensor([1, 2, 2]), topk=(1, 2))
    assert abs(p1.item() - 200 / 3) < 1e-4 and abs(p2.item() - 200 / 3) < 1e-4
