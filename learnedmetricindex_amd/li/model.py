"""MLP leaf predictor (reference: li/model.py).

Same public surface as the reference -- `ModelParameters`, `supported_models` (the nine `MLP..MLP-9`
variants, model.py:29-79), `Model`, `NeuralNetwork.{train,train_batch,predict,predict_proba}`,
`data_X_to_torch`, `data_to_torch`, `get_device`, `LIDataset` -- but inference (`predict`,
`predict_proba`) runs in the HIP kernels of liblmi_hip.so (lmi_mlp_topk / lmi_mlp_proba); torch only
stores the weights and runs the (offline, out-of-hot-path) training loop.
"""
from dataclasses import astuple, dataclass
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.utils.data
from torch import nn
from torch.nn import Linear, ReLU, Sequential

from .clustering import ClusteringAlgorithm
from .Logger import Logger

try:  # package import (learnedmetricindex_amd.li) or reference-style top-level `li`
    from .. import _capi
except ImportError:  # pragma: no cover
    import _capi  # type: ignore

torch.manual_seed(2023)
np.random.seed(2023)


@dataclass(frozen=True)
class ModelParameters:
    """Per-level hyper-parameters (model.py:17-27)."""

    clustering_algorithm: ClusteringAlgorithm
    model_type: str
    epochs: int
    lr: float
    n_categories: int

    def __iter__(self):
        return iter(astuple(self))


#: hidden widths of the reference's model zoo (model.py:29-79); "MLP-4" = in -> 512 -> out
_HIDDEN: Dict[str, Tuple[int, ...]] = {
    "MLP": (128,), "MLP-2": (64,), "MLP-3": (256,), "MLP-4": (512,), "MLP-5": (256, 128),
    "MLP-6": (32,), "MLP-7": (16,), "MLP-8": (8,), "MLP-9": (8, 16),
}


def _stack(hidden: Tuple[int, ...]) -> Callable[[int, int], Sequential]:
    def make(input_dim: int, output_dim: int) -> Sequential:
        mods: List[nn.Module] = []
        width = input_dim
        for hsize in hidden:
            mods += [Linear(width, hsize), ReLU()]
            width = hsize
        mods.append(Linear(width, output_dim))
        return Sequential(*mods)

    return make


supported_models: Dict[str, Callable[[int, int], Sequential]] = {k: _stack(v) for k, v in _HIDDEN.items()}


def init_layers(model_type: Optional[str], input_dim: int, output_dim: int) -> Sequential:
    if model_type not in supported_models:
        raise ValueError(f"Model type {model_type} not supported.")  # model.py:83-84
    return supported_models[model_type](input_dim, output_dim)


class Model(nn.Module):
    """`layers` is the Sequential the weights are read from (model.py:89-99)."""

    def __init__(self, input_dim=768, output_dim=1000, model_type: Optional[str] = None):
        super().__init__()
        self.layers = init_layers(model_type, input_dim, output_dim)
        self.n_output_neurons = output_dim

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.layers(x)


def data_X_to_torch(data) -> torch.Tensor:
    """np/pandas -> float32 torch tensor on the host (model.py:102-105)."""
    return torch.from_numpy(np.array(data).astype(np.float32))


def data_to_torch(data, labels) -> Tuple[torch.Tensor, torch.Tensor]:
    return data_X_to_torch(data), torch.as_tensor(torch.from_numpy(labels), dtype=torch.long)


def get_device() -> torch.device:
    """The current GPU (`cuda:0` unless the process selected another: one process per GPU in the
    sharded mode) when present, else cpu (model.py:115-127)."""
    if torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def linear_layers(module) -> List[Tuple[np.ndarray, np.ndarray]]:
    """[(W [out,in], b [out])] of every nn.Linear in a `Model` / Sequential, as float32 numpy."""
    seq = module.layers if hasattr(module, "layers") else module
    out = []
    for m in seq:
        if isinstance(m, nn.Linear):
            out.append((m.weight.detach().to("cpu", torch.float32).numpy(),
                        m.bias.detach().to("cpu", torch.float32).numpy()))
        elif not isinstance(m, nn.ReLU):
            raise ValueError(f"unsupported layer {type(m).__name__}: the MI355X path implements Linear/ReLU stacks")
    return out


def network_from_layers(layers, lr: float = 0.1) -> "NeuralNetwork":
    """A NeuralNetwork holding the given [(W [out,in], b [out])] stack: the matching `MLP*` variant
    when the hidden widths are one of the reference's, otherwise a custom Linear/ReLU Sequential."""
    dims = [layers[0][0].shape[1]] + [W.shape[0] for W, _ in layers]
    hidden = tuple(dims[1:-1])
    name = next((k for k, v in _HIDDEN.items() if v == hidden), None)
    net = NeuralNetwork(input_dim=dims[0], output_dim=dims[-1], lr=lr, model_type=name or "MLP")
    if name is None:
        net.model.layers = _stack(hidden)(dims[0], dims[-1]).to(net.device)
        net.optimizer = torch.optim.Adam(net.model.parameters(), lr=lr)
    lin = [m for m in net.model.layers if isinstance(m, nn.Linear)]
    with torch.no_grad():
        for m, (W, b) in zip(lin, layers):
            m.weight.copy_(torch.from_numpy(np.ascontiguousarray(W, dtype=np.float32)))
            m.bias.copy_(torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32)))
    return net


class NeuralNetwork(Logger):
    """One node's classifier (model.py:130-241).  Constructor arguments as in the reference."""

    _CHUNK = 1 << 18  # queries per HIP call in predict/predict_proba

    def __init__(self, input_dim, output_dim, loss=torch.nn.CrossEntropyLoss, lr=0.1, model_type="MLP",
                 class_weight=None):
        self.device = get_device()
        self.model = Model(input_dim, output_dim, model_type=model_type).to(self.device)
        self.loss = loss() if class_weight is None else loss(weight=class_weight.to(self.device))
        self.optimizer = torch.optim.Adam(self.model.parameters(), lr=lr)
        self._engine = None  # _capi.Index holding the packed weights; rebuilt after training

    # ---- pickling (search.py:234-241 pickles the whole index): the device handle is transient ----
    def __getstate__(self):
        state = dict(self.__dict__)
        state["_engine"] = None
        return state

    # ---- training: offline, torch autograd (out of the hot path) ----------
    def train(self, data_X: torch.Tensor, data_y: torch.Tensor, epochs=500, logger=None):
        """Full-batch Adam steps (model.py:161-183)."""
        self._engine = None
        self.model.train()
        step = max(epochs // 10, 1)
        X, y = data_X.to(self.device), data_y.to(self.device)
        losses = []
        for ep in range(epochs):
            loss = self.loss(self.model(X), y)
            if logger and ep and ep % step == 0:
                logger.debug(f"Epoch {ep} | Loss {loss.item()}")
            losses.append(loss.item())
            self.model.zero_grad()
            loss.backward()
            self.optimizer.step()
        return losses

    def train_batch(self, dataset, epochs=5, logger=None):
        """The reference's loop (model.py:185-211): every epoch runs the forward pass over all
        mini-batches but takes ONE optimizer step, on the last mini-batch's loss."""
        self._engine = None
        self.model.train()
        step = max(epochs // 10, 1)
        losses = []
        for ep in range(epochs):
            loss = None
            for bx, by in iter(dataset):
                loss = self.loss(self.model(bx.to(self.device)), by.to(self.device))
            if loss is None:
                break
            if logger and ep and ep % step == 0:
                logger.debug(f"Epoch {ep} | Loss {loss.item():.5f}")
            losses.append(loss.item())
            self.model.zero_grad()
            loss.backward()
            self.optimizer.step()
        return losses

    # ---- inference: HIP ---------------------------------------------------------------------
    def engine(self):
        """The device-side MLP (weights packed fragment-major in HBM)."""
        if self._engine is None:
            dev = self.device.index if self.device.type == "cuda" and self.device.index is not None else 0
            eng = _capi.Index(dev)
            eng.set_mlp(linear_layers(self.model))
            self._engine = eng
        return self._engine

    @staticmethod
    def _as_numpy(data_X) -> np.ndarray:
        if isinstance(data_X, torch.Tensor):
            data_X = data_X.detach().to("cpu", torch.float32).numpy()
        return np.ascontiguousarray(data_X, dtype=np.float32)

    def predict(self, data_X):
        """argmax class per row, int64 (model.py:213-224) -- used for object placement."""
        x = self._as_numpy(data_X)
        eng = self.engine()
        out = np.empty(x.shape[0], dtype=np.int64)
        for r0 in range(0, x.shape[0], self._CHUNK):
            out[r0: r0 + self._CHUNK] = eng.mlp_topk(x[r0: r0 + self._CHUNK], 1)[:, 0]
        return out

    def predict_proba(self, data_X):
        """(probs f32[n,L] descending, classes i64[n,L]) (model.py:226-241)."""
        x = self._as_numpy(data_X)
        assert x.ndim == 2, "predict_proba expects a 2-D batch (the reference's 1-D path is dead code)"
        eng = self.engine()
        L = eng.n_classes
        probs = np.empty((x.shape[0], L), dtype=np.float32)
        classes = np.empty((x.shape[0], L), dtype=np.int64)
        for r0 in range(0, x.shape[0], self._CHUNK):
            p, c = eng.mlp_proba(x[r0: r0 + self._CHUNK])
            probs[r0: r0 + self._CHUNK], classes[r0: r0 + self._CHUNK] = p, c
        return probs, classes


class LIDataset(torch.utils.data.Dataset):
    """1-based indexable (x, y) pairs for SubsetRandomSampler over DataFrame labels (model.py:244-252)."""

    def __init__(self, dataset_x, dataset_y):
        self.dataset_x, self.dataset_y = data_to_torch(dataset_x, dataset_y)

    def __len__(self):
        return self.dataset_x.shape[0]

    def __getitem__(self, idx):
        return self.dataset_x[idx - 1], self.dataset_y[idx - 1]
