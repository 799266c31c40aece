"""Parameter containers with the reference's module names / shapes / init.
Only construction lives here; execution is in layers.py (HIP)."""
import torch.nn as nn


def init_small(m):
    """N(0,1e-3) weights, zero bias; BN2d to (1,0)   (nets/net_utils.py:22-33)"""
    if isinstance(m, (nn.Conv2d, nn.Linear, nn.ConvTranspose2d)):
        m.weight.data.normal_(0, 1e-3)
        if m.bias is not None:
            m.bias.data.zero_()
    elif isinstance(m, nn.BatchNorm2d):
        m.weight.data.fill_(1)
        m.bias.data.zero_()


def conv_1x1(cin, cout, leaky):
    """Conv1d(k=1)+ReLU/LeakyReLU(0.1)  (net_utils.py:35-43; Conv1d keeps torch's default init)"""
    act = nn.LeakyReLU(0.1, inplace=True) if leaky else nn.ReLU(inplace=True)
    return nn.Sequential(nn.Conv1d(cin, cout, 1, 1, 0, bias=True), act)


def conv_bn_relu(cin, cout, k, stride=1, padding=0):
    seq = nn.Sequential(nn.Conv2d(cin, cout, k, stride, padding, bias=False), nn.BatchNorm2d(cout),
                        nn.LeakyReLU(0.2, inplace=True))
    seq.apply(init_small)
    return seq


def convt_bn_relu(cin, cout, k, stride=1, padding=0, output_padding=0):
    seq = nn.Sequential(nn.ConvTranspose2d(cin, cout, k, stride, padding, output_padding, bias=False),
                        nn.BatchNorm2d(cout), nn.LeakyReLU(0.2, inplace=True),
                        nn.Conv2d(cout, cout, 3, 1, 1, bias=False), nn.BatchNorm2d(cout),
                        nn.LeakyReLU(0.2, inplace=True))
    seq.apply(init_small)
    return seq


class VGGFeatures(nn.Module):
    """`features` of nets/vgg.py:69-83 with kaiming-normal(fan_out) conv init (:55-66)."""
    CFG = {'A': [64, 'M', 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'],
           'C': [64, 'M', 128, 'M', 256, 256, 'M', 512, 512, 'M']}

    def __init__(self, cfg):
        super().__init__()
        layers, cin = [], 3
        for v in self.CFG[cfg]:
            if v == 'M':
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(cin, v, 3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)


class BasicBlock(nn.Module):
    """parameter layout of nets/resnet.py:33-53"""

    def __init__(self, inplanes, planes, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False),
                                            nn.BatchNorm2d(planes))


def resnet18_layers():
    """layer1..4 of resnet18 (resnet.py:171-193, 226-234), re-initialised N(0,1e-3) (gnet.py:32,83)"""
    out, inpl = [], 64
    for planes, stride in ((64, 1), (128, 2), (256, 2), (512, 2)):
        layer = nn.Sequential(BasicBlock(inpl, planes, stride), BasicBlock(planes, planes, 1))
        layer.apply(init_small)
        out.append(layer)
        inpl = planes
    return out


class BilateralConvFlex(nn.Module):
    """parameter layout of nets/bilateralNN.py:55-139 for do_splat=True, do_slice=False, two outputs"""

    def __init__(self, num_input, num_output):
        super().__init__()
        import torch
        self.num_input, self.num_output = num_input, list(num_output)
        self.register_buffer('feat_indices', torch.arange(num_input, dtype=torch.long))
        self.blur_conv = nn.Sequential(nn.Conv2d(num_input, num_output[0], (15, 1), 1, 0, bias=True),
                                       nn.ReLU(inplace=False),
                                       nn.Conv2d(num_output[0], num_output[1], (1, 1)))
        self.blur_conv.apply(init_small)
