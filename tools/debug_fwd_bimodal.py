import copy, sys, torch
sys.path.insert(0, '/root/repo')
from epipolarpose_amd import hip
from epipolarpose_amd.core.config import default_config
from epipolarpose_amd.models.pose3d_resnet import get_pose_net, deconv_bn_act
from epipolarpose_amd.optim import FusedAdam
dev = torch.device("cuda:0")
cfg = default_config(); cfg.MODEL.INIT_WEIGHTS = False; cfg.MODEL.EXTRA.NUM_LAYERS = 18
j = 4
cfg.MODEL.NUM_JOINTS, cfg.MODEL.DEPTH_RES, cfg.MODEL.IMAGE_SIZE = j, 16, [64, 64]
torch.manual_seed(2)
base = get_pose_net(cfg, is_train=False).to(dev).train()
x = torch.randn(8, 3, 64, 64, device=dev).contiguous(memory_format=torch.channels_last)
def run():
    m = copy.deepcopy(base)
    opt = FusedAdam(m, lr=1e-2)
    outs = []
    with torch.no_grad() if False else torch.enable_grad():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            t, _ = m.stem(x); outs.append(("stem", t))
            t = hip.glue().maxpool3x3s2(t); outs.append(("pool", t))
            for ln in ("layer1", "layer2", "layer3", "layer4"):
                for bi, blk in enumerate(getattr(m, ln)):
                    t = blk(t); outs.append(("%s.%d" % (ln, bi), t))
            for pi, (deconv, bn) in enumerate(m._head_pairs):
                t = deconv_bn_act(deconv, bn, t); outs.append(("deconv%d" % pi, t))
            t = m.final_layer(t); outs.append(("final", t))
    torch.cuda.synchronize()
    return [(n, v.detach().float().clone()) for n, v in outs]
ref = run()
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    cur = run()
    line = []
    for (n, a), (_, b) in zip(ref, cur):
        d = float((a - b).abs().max() / a.abs().max())
        line.append("%s %.1e" % (n, d))
    print(it, " ".join(line), flush=True)
