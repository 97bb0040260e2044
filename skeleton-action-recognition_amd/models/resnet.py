"""Drop-in for the reference's models/resnet.py: `Model(num_classes, num_filters, image_size, device)` =
VirtualRadar(wavelength=5e-4) -> (B,1,n_fft,F) -> nearest resize to image_size -> ResNet-18 (models/resnet.py:11-28).
The resize is fused into the STFT kernel as a column select when n_fft == image_size (the reference's default
256/256: rows are copied, columns are picked by floor(j*F/256))."""
import torch

from layers.virtual_radar import VirtualRadar
from models.resnet18 import resnet18


class Model(torch.nn.Module):
    def __init__(self, num_classes=60, num_filters=64, image_size=256, device='cuda:0', num_pad_frames=0, sigma=3, mfma=None):
        """num_pad_frames = P > 0 (not in the reference's constructor): feed RAW clips and let the radar layer apply
        utils.Dataset.pad_frames (smoothing + x P cubic up-sampling, the reference's CPU data-loader step) on the GPU."""
        super().__init__()
        self.num_pad_frames, self.sigma = num_pad_frames, sigma
        self.base_model = resnet18(num_classes=num_classes, num_filters=num_filters, device=device, mfma=mfma)      # mfma: models/resnet18.py
        self.virtual_radar = VirtualRadar(wavelength=5e-4, device=device)
        self.image_size = image_size

    def spectrogram(self, x):
        if self.virtual_radar.n_fft == self.image_size:
            return self.virtual_radar(x, out_cols=self.image_size, num_pad_frames=self.num_pad_frames,
                                      sigma=self.sigma).unsqueeze(1)
        s = self.virtual_radar(x, num_pad_frames=self.num_pad_frames, sigma=self.sigma).unsqueeze(1)
        return torch.nn.functional.interpolate(s, self.image_size)

    def forward(self, x):
        return self.base_model(self.spectrogram(x))
