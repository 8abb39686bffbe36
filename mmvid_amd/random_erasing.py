"""RandomErasing box sampler.  The reference uses torchvision.transforms.RandomErasing (dalle_bert.py:292-295,
427-432; dalle_artv.py:229-232), a third-party op (torchvision is not installed here): PARITY UNPINNED, restated
from the upstream semantics -- area ~ U(scale)*H*W, log-uniform aspect ratio, up to 10 tries, box applied to the
last two dims of every leading index (so a [T,1,h,w] mask gets the same box on all frames)."""
import math

import torch


class RandomErasing(torch.nn.Module):
    def __init__(self, p=0.5, scale=(0.02, 0.33), ratio=(0.3, 3.3), value=0, inplace=False):
        super().__init__()
        self.p, self.scale, self.ratio, self.value, self.inplace = p, scale, ratio, value, inplace

    @staticmethod
    def get_params(img, scale, ratio):
        img_h, img_w = img.shape[-2], img.shape[-1]
        area = img_h * img_w
        log_ratio = torch.log(torch.tensor(ratio))
        for _ in range(10):
            erase_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
            aspect_ratio = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
            h = int(round(math.sqrt(erase_area * aspect_ratio)))
            w = int(round(math.sqrt(erase_area / aspect_ratio)))
            if not (h < img_h and w < img_w):
                continue
            i = torch.randint(0, img_h - h + 1, size=(1, )).item()
            j = torch.randint(0, img_w - w + 1, size=(1, )).item()
            return i, j, h, w
        return 0, 0, img_h, img_w

    def forward(self, img):
        if torch.rand(1) < self.p:
            i, j, h, w = self.get_params(img, self.scale, self.ratio)
            if (i, j, h, w) == (0, 0, img.shape[-2], img.shape[-1]):
                return img  # upstream returns the original image when no box was found
            if not self.inplace:
                img = img.clone()
            img[..., i:i + h, j:j + w] = self.value
        return img
