"""``NeuralModuleNetwork`` -- class surface of the reference's ``probnmn.models.nmn`` (reference:
probnmn/models/nmn.py:23-296) executed by the gfx950 engine.

Same constructor / ``from_config`` arguments, same submodule and ``state_dict`` names (``stem.0``,
``stem.2``, ``classifier.{0,4,6}``, one child per program token), same ``forward`` signature and
return dict.  What differs is how ``forward`` gets there: programs are compiled statically
(validity included -- the reference's bare ``except`` is not reproduced, kernel errors surface),
all module calls of the batch run as grouped kernels, and stem / classifier conv / max-pool are
part of the same explicit forward+backward schedule (``probnmn.runtime.engine``).
"""
from typing import Dict, Optional, Tuple

import torch
from torch import nn
from torch.nn import functional as F

from probnmn.modules.nmn_modules import (
    AndModule,
    AttentionModule,
    ComparisonModule,
    Flatten,
    OrModule,
    QueryModule,
    RelateModule,
    SameModule,
)
from probnmn.runtime import program_compiler as pc
from probnmn.running_metrics import Average, BooleanAccuracy

INVALID_PROGRAM_LOSS = 3.33  # ~ ln(28), the reference's constant (nmn.py:260,269)


class _Tokens:
    """A batch of programs as the host token matrix, on its way through the library's trunk planner
    (``NMNEngine.run_forward_tokens``); ``valid`` is filled in by the forward pass."""

    __slots__ = ("tokens", "valid")

    def __init__(self, tokens):
        self.tokens, self.valid = tokens, None


class _Trunk(torch.autograd.Function):
    """stem -> module programs -> classifier conv + max-pool, as one autograd node."""

    @staticmethod
    def forward(ctx, features, engine, compiled, started, *params):
        need_backward = any(ctx.needs_input_grad)  # false under torch.no_grad()
        # the library compiles, plans and launches from the token matrix
        pooled, state, compiled.valid = engine.run_forward_tokens(features, compiled.tokens, need_backward, started)
        ctx.engine, ctx.state, ctx.n_params = engine, state, len(params)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        if ctx.state is None:
            raise RuntimeError("backward through a forward that was run without gradient tracking")
        grads = ctx.engine.run_backward(ctx.state, dpooled)
        return (None, None, None, None, *grads[: ctx.n_params])


class _SplitKLinear(torch.autograd.Function):
    """``F.linear`` for the classifier's first fully connected layer (K = class_projection_channels * H * W / 4 = 50 176
    inputs -> 1024 units; reference nmn.py:76-83) with the FORWARD product split along K: x [B, K] @ W^T [K, N] has only
    B/32 x N/96 output tiles for the library GEMM (22 workgroups at 64 rows, 170 at 512) over a 50 176-long reduction;
    sixteen K-slabs as ONE strided-batched GEMM (no copies: both operands are views) fill the chip and the partial
    products add up after.  Measured (scripts/fc_gemm_probe.py): 179 -> 65 us at 64 rows, 579 -> 398 us at 512.
    The two gradient products already have large outputs and stay plain GEMMs."""

    SLABS = 16

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        B, K = x.shape
        N, S = weight.size(0), _SplitKLinear.SLABS
        xs = x.view(B, S, K // S).transpose(0, 1)             # [S, B, K/S]
        ws = weight.view(N, S, K // S).permute(1, 2, 0)       # [S, K/S, N]
        return torch.bmm(xs, ws).sum(0).add_(bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx = dy @ weight if ctx.needs_input_grad[0] else None
        dw = dy.t() @ x if ctx.needs_input_grad[1] else None
        db = dy.sum(0) if ctx.needs_input_grad[2] else None
        return dx, dw, db


class _WeightGradOnly(torch.autograd.Function):
    """The weight / bias gradient of a linear layer as a node of its OWN: forward contributes zeros, backward computes
    dy^T x and sum(dy).  With ``_first_fc`` below the layer's backward is two nodes -- d(input) first, then this one -- so
    that what waits for d(input) (the trunk's backward, on its own stream) is released by the event behind the first GEMM
    instead of behind both: the 50 176 x 1024 weight-gradient GEMM (0.45 ms at 512 rows) then runs beside the trunk's first
    backward launches instead of in front of them.  (Deferring it to the end of backward instead would also delay the
    early all-reduce of this 205 MB gradient under data parallelism.)"""

    @staticmethod
    def forward(ctx, weight, bias, x):
        ctx.save_for_backward(x)
        return x.new_zeros(x.size(0), weight.size(0))

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return (dy.t() @ x if ctx.needs_input_grad[0] else None, dy.sum(0) if ctx.needs_input_grad[1] else None, None)


# ---- the same layer on this build's GEMM (csrc/gemm.hip) ----------------------------------------------------------------
# OPT-IN (PNMN_FC_OWN_ROWS=<rows>: from that many rows on; default 0 = never): the three products of the layer on pnmn_gemm --
# whole 128-row tiles straight into LDS, the forward product split along K so that its 8 x rows/128 output tiles fill the
# chip (partials added in chunk order: deterministic), the bias added in the epilogue, the bias gradient summed beside the
# weight gradient (colsum).  Measured at 512 rows (the 1024-question step): forward 503 us, d(input) 499, weight gradient
# 460 against the library path's 398 / 437 / 418 -- the step 27.60 against 27.13 ms (28.99 / 28.59 on one stream); at 64 rows
# a 128-row tile is half empty (41 against 88 TFLOP/s).  So the library path above stays the default (DESIGN 4.2 "Round 6").
OWN_FC_ROWS = int(__import__("os").environ.get("PNMN_FC_OWN_ROWS", "0"))
_FC_WORKSPACES: dict = {}


def _own_gemm(a, b, c, M, N, K, lda, ldb, ldc, ta=False, tb=False, bias=None, colsum=None, split=1):
    import numpy as np
    from probnmn import _hip

    lib, dev = _hip.lib(), c.device
    d = np.zeros(1, _hip.GEMM_DESC)
    d["a"], d["b"], d["c"] = a.data_ptr(), b.data_ptr(), c.data_ptr()
    d["lda"], d["ldb"], d["ldc"], d["M"], d["N"], d["K"] = lda, ldb, ldc, M, N, K
    d["flags"] = (_hip.GEMM_A_T if ta else 0) | (_hip.GEMM_B_T if tb else 0)
    d["split_k"] = split
    if bias is not None:
        d["bias"] = bias.data_ptr()
    if colsum is not None:
        d["colsum"] = colsum.data_ptr()
    if split > 1:
        need = int(lib.pnmn_gemm_workspace_bytes(M, N, split))
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        ws = _FC_WORKSPACES.get(key)
        if ws is None or ws.numel() < need:  # (one grow-only buffer per device: the products of a step run one after another)
            ws = _FC_WORKSPACES[key] = torch.empty(need, dtype=torch.uint8, device=dev)
        d["workspace"] = ws.data_ptr()
    _hip.check(lib.pnmn_gemm(d.ctypes.data, 1, _hip.stream_ptr(dev)), "pnmn_gemm (fully connected layer)")


class _OwnLinear(torch.autograd.Function):
    """``F.linear`` of the first fully connected layer on pnmn_gemm: y = x W^T + b; backward: d(input) only (the weight
    gradient is ``_OwnWeightGrad``'s, see ``_WeightGradOnly`` for why the two are separate nodes)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from probnmn import _hip

        ctx.save_for_backward(weight)
        M, K = x.shape
        N = weight.size(0)
        y = torch.empty(M, N, dtype=x.dtype, device=x.device)
        split = int(_hip.lib().pnmn_gemm_split_k(M, N, K, 0))
        _own_gemm(x, weight, y, M, N, K, K, K, N, tb=True, bias=bias, split=split)
        return y

    @staticmethod
    def backward(ctx, dy):
        (weight,) = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None, None
        dy = dy.contiguous()
        M, N = dy.shape
        K = weight.size(1)
        dx = torch.empty(M, K, dtype=dy.dtype, device=dy.device)
        _own_gemm(dy, weight, dx, M, K, N, N, K, K)  # dy [M][N] . W [N][K]
        return dx, None, None


class _OwnWeightGrad(torch.autograd.Function):
    """``_WeightGradOnly`` on pnmn_gemm: dW = dy^T x with db = the column sums of dy from the same pass."""

    @staticmethod
    def forward(ctx, weight, bias, x):
        ctx.save_for_backward(x)
        return x.new_zeros(x.size(0), weight.size(0))

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        R, N = dy.shape
        K = x.size(1)
        dw = torch.empty(N, K, dtype=dy.dtype, device=dy.device)
        db = torch.empty(N, dtype=dy.dtype, device=dy.device)
        _own_gemm(dy, x, dw, N, K, R, N, K, K, ta=True, colsum=db)  # dy^T [N][R] . x [R][K]
        return dw, db, None


def _first_fc(layer: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    K = layer.in_features
    if (x.is_cuda and x.dim() == 2 and x.is_contiguous() and layer.weight.is_contiguous() and layer.bias is not None
            and x.dtype == torch.float32 and 0 < OWN_FC_ROWS <= x.size(0) and K % 4 == 0):
        if torch.is_grad_enabled() and x.requires_grad and layer.weight.requires_grad:
            from_weights = _OwnWeightGrad.apply(layer.weight, layer.bias, x.detach())
            return _OwnLinear.apply(x, layer.weight.detach(), layer.bias.detach()) + from_weights
        if not torch.is_grad_enabled() or not (x.requires_grad or layer.weight.requires_grad or layer.bias.requires_grad):
            return _OwnLinear.apply(x, layer.weight, layer.bias)
        # (gradients for some of the three only: the library path below knows every combination)
    if (x.is_cuda and x.dim() == 2 and x.is_contiguous() and layer.weight.is_contiguous() and layer.bias is not None
            and K % _SplitKLinear.SLABS == 0 and K >= 8192):
        if torch.is_grad_enabled() and x.requires_grad and layer.weight.requires_grad:
            # (autograd runs the node created LAST first: the weight-gradient node is made before the product)
            from_weights = _WeightGradOnly.apply(layer.weight, layer.bias, x.detach())
            return _SplitKLinear.apply(x, layer.weight.detach(), layer.bias.detach()) + from_weights
        return _SplitKLinear.apply(x, layer.weight, layer.bias)
    return layer(x)


class _AnswerLoss(torch.autograd.Function):
    """``pnmn_answer_loss``: log-softmax over the answers, arg-max prediction, cross entropy, the
    invalid-program overrides (prediction @@UNKNOWN@@, constant loss 3.33, no gradient) and d loss / d logits
    in one launch (reference nmn.py:245-269)."""

    @staticmethod
    def forward(ctx, logits, answers, valid, unknown_index):
        from probnmn import _hip

        logits = logits.contiguous()
        B, A = logits.shape
        dev = logits.device
        predictions = torch.empty(B, dtype=torch.long, device=dev)
        loss = torch.empty(B, dtype=torch.float32, device=dev)
        dlogits = torch.empty(B, A, dtype=torch.float32, device=dev) if answers is not None else None
        _hip.check(_hip.lib().pnmn_answer_loss(
            logits.data_ptr(), 0 if answers is None else answers.data_ptr(), valid.data_ptr(), predictions.data_ptr(),
            loss.data_ptr(), 0 if dlogits is None else dlogits.data_ptr(), B, A, unknown_index, 1.0,
            _hip.stream_ptr(dev)), "answer_loss")
        ctx.save_for_backward(dlogits)
        ctx.mark_non_differentiable(predictions)
        return loss, predictions

    @staticmethod
    def backward(ctx, dloss, _):
        (dlogits,) = ctx.saved_tensors
        if dlogits is None:
            raise RuntimeError("the answer loss without answers (-max log-probability) is an evaluation quantity")
        return dlogits * dloss.unsqueeze(1), None, None, None


class NeuralModuleNetwork(nn.Module):
    def __init__(
        self,
        vocabulary,
        image_feature_size: Tuple[int, int, int] = (1024, 14, 14),
        module_channels: int = 128,
        class_projection_channels: int = 1024,
        classifier_linear_size: int = 1024,
    ):
        super().__init__()
        self.vocabulary = vocabulary
        channels, height, width = image_feature_size
        # "@@UNKNOWN@@" is never produced by the classifier (reference nmn.py:60-63)
        num_answers = len(vocabulary.get_index_to_token_vocabulary(namespace="answers")) - 1

        self.stem = nn.Sequential(
            nn.Conv2d(channels, module_channels, kernel_size=3, padding=1),
            nn.ReLU(),
            nn.Conv2d(module_channels, module_channels, kernel_size=3, padding=1),
            nn.ReLU(),
        )
        self.classifier = nn.Sequential(
            nn.Conv2d(module_channels, class_projection_channels, kernel_size=1),
            nn.ReLU(),
            nn.MaxPool2d(kernel_size=2, stride=2),
            Flatten(),
            nn.Linear(class_projection_channels * height * width // 4, classifier_linear_size),
            nn.ReLU(),
            nn.Linear(classifier_linear_size, num_answers),
        )

        # one child module per program token, named by the token (reference nmn.py:86-115)
        self._function_modules: Dict[str, Optional[nn.Module]] = {}
        factories = {pc.AND: AndModule, pc.OR: OrModule}
        sized = {pc.CMP: ComparisonModule, pc.QUERY: QueryModule, pc.REL: RelateModule,
                 pc.SAME: SameModule, pc.ATT: AttentionModule}
        for token in vocabulary.get_token_to_index_vocabulary("programs"):
            kind = pc.classify_token(token)
            if kind == pc.SKIP:
                continue
            if kind == pc.SCENE:
                module = None
            elif kind in factories:
                module = factories[kind]()
            else:
                module = sized[kind](module_channels)
            self._function_modules[token] = module
            self.add_module(token, module)

        self._unknown_answer = vocabulary.get_token_index("@@UNKNOWN@@", namespace="answers")
        self._answer_accuracy = BooleanAccuracy()
        self._average_invalid_programs = Average()
        # reference behaviour: every training forward returns batch metrics as Python floats, which
        # costs a device->host sync per step; trainers that log less often switch this off
        self.report_batch_metrics = True

        from probnmn.runtime.engine import NMNEngine

        self._engine = NMNEngine(self, tuple(image_feature_size), module_channels, class_projection_channels)

    @classmethod
    def from_config(cls, config):
        from probnmn.vocabulary import Vocabulary

        _C = config
        return cls(
            vocabulary=Vocabulary.from_files(_C.DATA.VOCABULARY),
            image_feature_size=tuple(_C.NMN.IMAGE_FEATURE_SIZE),
            module_channels=_C.NMN.MODULE_CHANNELS,
            class_projection_channels=_C.NMN.CLASS_PROJECTION_CHANNELS,
            classifier_linear_size=_C.NMN.CLASSIFIER_LINEAR_SIZE,
        )

    @property
    def engine(self):
        return self._engine

    def forward(self, features: torch.Tensor, programs: torch.Tensor, answers: Optional[torch.Tensor] = None,
                started=None, trunk_stream=None, rows: Optional[torch.Tensor] = None):
        # ``started``: token of ``begin(features)`` when the caller already launched the stem (optional)
        # ``rows``: run on ``features[rows]`` (programs / answers belong to those examples) without the gathered copy
        # ``trunk_stream``: run the trunk (stem, module programs, classifier conv + pool -- this build's own
        # kernels, none of which waits for another workgroup) on that stream, forward and backward, beside
        # whatever the caller queues on the current stream; the fully connected layers and the loss stay on
        # the current stream (see DESIGN 6 for why the library GEMMs must not leave it)
        return self.forward_head(self.forward_trunk(features, programs, started, trunk_stream, rows), answers)

    def forward_trunk(self, features: torch.Tensor, programs: torch.Tensor, started=None, trunk_stream=None,
                      rows: Optional[torch.Tensor] = None):
        """First half of ``forward``: compile and schedule the programs, launch the trunk (on ``trunk_stream`` when
        given).  Nothing is queued on the current stream, so a caller that runs the trunk on its own stream can
        keep feeding the current one before ``forward_head`` makes it wait for the trunk."""
        engine = self._engine
        arena = engine.ensure_arena()
        if rows is not None and started is None:
            started = self.begin(features, rows)
        # the programs decide the launch schedule, so they are needed on the host (the reference
        # also reads them back, once per example: nmn.py:203)
        # (a CPU ``programs`` tensor costs nothing; a device tensor costs one device->host sync)
        tokens = programs.detach().cpu().numpy()
        compiled = _Tokens(tokens)

        # the trunk's parameters as inputs of its autograd node -- all of them when autograd is to receive
        # their gradients; ONE anchor when a trainer reads the gradients straight from the arena
        # (engine.direct_grads): 218 inputs cost ~0.5 ms of host time per step in apply() and in 218
        # AccumulateGrad visits that carry nothing
        params = [arena.param(n) for n in arena.names]
        if engine.direct_grads:
            params = params[:1]
        if trunk_stream is not None:
            with torch.cuda.stream(trunk_stream):
                pooled = _Trunk.apply(features, engine, compiled, started, *params)
        else:
            pooled = _Trunk.apply(features, engine, compiled, started, *params)
        return pooled, compiled, trunk_stream

    def forward_head(self, trunk, answers: Optional[torch.Tensor] = None):
        """Second half of ``forward``: the fully connected layers and the loss, on the current stream."""
        from probnmn import _hip

        pooled, compiled, trunk_stream = trunk
        # (staged here, not in forward_trunk: the host's time between the sampled programs' arrival and the trunk's
        # launch is on the critical path of a small-batch step)
        valid_host = compiled.valid.tolist()
        valid = _hip.small_to_device(valid_host, torch.int32, pooled.device)
        self._engine.wait_params(pooled.device)  # (an optimiser step of the FC layers still running on the trunk's stream)
        if trunk_stream is not None:
            current = torch.cuda.current_stream(pooled.device)
            current.wait_stream(trunk_stream)
            pooled.record_stream(current)
        hidden = F.relu(_first_fc(self.classifier[4], pooled))
        answer_logits = self.classifier[6](hidden)
        _hip.mark("classifier FC forward done")
        if answer_logits.requires_grad:
            answer_logits.register_hook(lambda g: _hip.mark("d(answer logits) arrives"))
            pooled.register_hook(lambda g: _hip.mark("d(pooled) computed (FC backward done)"))

        # log-softmax, arg-max, cross entropy (or -max log-probability without answers) and the
        # invalid-program overrides -- prediction @@UNKNOWN@@, constant loss 3.33, no gradient -- in one kernel
        if answers is not None:
            answers = answers.contiguous()
        loss, answer_predictions = _AnswerLoss.apply(answer_logits, answers, valid, self._unknown_answer)
        if answers is not None and (self.report_batch_metrics or not self.training):
            self._answer_accuracy(answer_predictions, answers)
            self._average_invalid_programs(sum(1 for v in valid_host if not v))

        output_dict = {"predictions": answer_predictions, "loss": loss}
        if self.training and self.report_batch_metrics:
            output_dict["metrics"] = self.get_metrics(reset=True)
        return output_dict

    def begin(self, features: torch.Tensor, rows: Optional[torch.Tensor] = None):
        """Launch the program-independent part of ``forward`` (feature layout + stem) ahead of time;
        pass the returned token as ``forward(..., started=token)`` with the same ``features``.  ``rows`` (int64
        device tensor): the pass runs on ``features[rows]`` -- programs / answers of ``forward`` then belong to
        those examples -- without the gathered copy of the feature maps."""
        trains = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        return self._engine.begin_forward(features, trains, rows)

    def get_metrics(self, reset: bool = True) -> Dict[str, float]:
        return {
            "answer_accuracy": self._answer_accuracy.get_metric(reset=reset),
            "average_invalid": self._average_invalid_programs.get_metric(reset=reset),
        }
