"""Small stand-ins for the two Keras `Model` objects the reference's builders return."""
import numpy as np
import torch


class TrainingModelStub:
    """The reference's first return value (the loss-producing training model).
    Training is outside this inference path."""

    name = 'f_pointnet_train'

    def __call__(self, *a, **k):
        raise NotImplementedError("training (losses, gradients, optimiser) is outside this inference path")

    fit = fit_generator = __call__


class DetectionModel:
    """The reference's `det_model`: inputs [frustum_point_cloud, one_hot_vec], outputs
    [logits, center, heading_scores, heading_residual, size_scores, size_residual]
    (models/FPointNet_Keras.py:187-189, models/FPointNet_Bayes.py:219-221).
    `predict` takes/returns numpy like Keras (a dict keyed by the input names or a
    list); calling the object takes/returns device tensors."""

    name = 'f_pointnet_inference'
    OUTPUTS = ("mask_logits", "center", "heading_scores", "heading_residuals", "size_scores", "size_residuals")

    def __init__(self, forward, store, device):
        self._forward = forward
        self.store = store
        self.device = device

    def __call__(self, point_cloud, one_hot_vec):
        ep = self._forward(point_cloud.contiguous(), one_hot_vec.contiguous())
        return [ep[k] for k in self.OUTPUTS]

    def predict(self, x, batch_size=None, steps=None):
        if isinstance(x, dict):
            pc, oh = x["frustum_point_cloud"], x["one_hot_vec"]
        else:
            pc, oh = x[0], x[1]
        pc = torch.as_tensor(np.ascontiguousarray(pc, dtype=np.float32)).to(self.device)
        oh = torch.as_tensor(np.ascontiguousarray(oh, dtype=np.float32)).to(self.device)
        return [t.cpu().numpy() for t in self(pc, oh)]

    def detect(self, point_cloud, one_hot_vec, rot_angle=None, want_seg=False):
        """Network + detection-record kernel, nothing but the records leaves the device:
        inputs as numpy arrays or device tensors, rot_angle (B,) or None ->
        (records (B,20) float64, seg_bits (B,ceil(N/32)) int32 or None) device tensors
        (fpn.inference.detection_records; scripts/test.py:145-160 + provider.py:366-373)."""
        from .. import inference

        def dev(a):
            if isinstance(a, torch.Tensor):
                return a.to(self.device, torch.float32).contiguous()
            return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)
        ep = self._forward(dev(point_cloud), dev(one_hot_vec))
        rot = None if rot_angle is None else dev(rot_angle).reshape(-1)
        return inference.detection_records(ep["mask_logits"].contiguous(), ep["box_output"].contiguous(),
                                           ep["stage1_center"].contiguous(), rot, want_seg=want_seg)

    def load_weights(self, mapping):
        """mapping: variable name -> array (see fpn.variables for the names)."""
        self.store.load(mapping)
        return self
