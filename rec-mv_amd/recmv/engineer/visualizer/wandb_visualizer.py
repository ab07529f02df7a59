"""`wandb_visualizer(project_name, exp_name, resume)` of the reference (engineer/visualizer/wandb_visualizer.py:7-60), optional:
with Weights & Biases installed and not disabled (`WANDB_MODE=disabled`) scalars and images go there exactly as in the reference;
otherwise scalars are appended to `<log_dir>/<exp_name>.jsonl` (one JSON object per call, `step` included) and images are
dropped.  Either way the object has the methods the optimisation code calls: `add_scalar(dict, step)`, `add_image(dict, step,
size, rgb, normalized)`, `watch_model(model)`."""
import datetime
import json
import os


class wandb_visualizer:
    def __init__(self, project_name, exp_name, resume=False, log_dir='./logs'):
        self.run = None
        self.path = None
        stamp = datetime.datetime.now().strftime("%Y-%m-%d-%H-%M")
        if os.environ.get('WANDB_MODE', '') != 'disabled':
            try:
                import wandb
                self.run = wandb.init(project=project_name, name=exp_name + '_' + stamp, dir=log_dir, resume=resume)
                self._wandb = wandb
            except Exception:          # not installed, or no network / credentials: fall back to the file
                self.run = None
        if self.run is None:
            os.makedirs(log_dir, exist_ok=True)
            self.path = os.path.join(log_dir, '%s.jsonl' % exp_name)

    def watch_model(self, model):
        if self.run is not None:
            self._wandb.watch(model)

    def add_scalar(self, scalar_dict, step):
        clean = {k: float(v) for k, v in scalar_dict.items()}
        if self.run is not None:
            self._wandb.log(clean, step)
            return
        with open(self.path, 'a') as fh:
            fh.write(json.dumps(dict(clean, step=int(step))) + '\n')

    def add_image(self, tensor_dict, step, size=256, rgb=True, normalized=False):
        if self.run is None:
            return
        import numpy as np
        out = {}
        for k, t in tensor_dict.items():
            a = t.detach().float().cpu().numpy()
            if normalized:
                a = (a / 2. + 0.5) * 255.
            a = np.clip(a, 0, 255).astype(np.uint8)
            out[k] = self._wandb.Image(a[..., ::-1] if rgb and a.ndim == 3 and a.shape[-1] == 3 else a)
        self._wandb.log(out, step)
