"""Feature ingest: pinned host store -> one gather kernel -> NHWC device batch, used by the
network in place; prefetching loader.  Reference behaviour being replaced: readers.py:63-108 (row lookup),
datasets.py:137-142 (float cast), _trainer.py:272-287 (.to(device))."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("size,dtype", [(14, np.float64), (28, np.float32)])
def test_gather_equals_host_indexing(size, dtype):
    from probnmn.data.feature_store import PinnedFeatureStore

    rng = np.random.Generator(np.random.Philox(size))
    feats = rng.standard_normal((37, 1024, size, size)).astype(dtype)  # h5 features are float64 (extract_features.py:119-121)
    store = PinnedFeatureStore(feats, chunk_rows=10)
    assert len(store) == 37 and store.image_feature_size == (1024, size, size)
    idx = torch.tensor([5, 0, 36, 5, 17, 22, 1])
    got = store.gather(idx, DEV)
    assert got.shape == (7, 1024, size, size) and got.is_contiguous(memory_format=torch.channels_last)
    want = torch.from_numpy(feats[idx.numpy()]).float()  # the reference's lookup + cast
    assert torch.equal(got.cpu(), want)
    with pytest.raises(IndexError):
        store.gather(torch.tensor([37]), DEV)
    # indices already on the device
    assert torch.equal(store.gather(idx.to(DEV), DEV).cpu(), want)


def test_network_takes_the_gathered_batch_in_place():
    """A channels_last batch (what the store produces) goes through the network without a layout pass and
    gives bit-identical outputs and gradients to the same batch handed over as contiguous NCHW."""
    from probnmn.data.feature_store import PinnedFeatureStore
    from probnmn.data.synthetic import synthetic_batch
    from probnmn.models import NeuralModuleNetwork
    from probnmn.vocabulary import Vocabulary

    vocab = Vocabulary.clevr()
    torch.manual_seed(0)
    nmn = NeuralModuleNetwork(vocab, class_projection_channels=128, classifier_linear_size=64).to(DEV)
    nmn.train()
    b = synthetic_batch(vocab, 12, seed=4)
    store = PinnedFeatureStore(b["image"].numpy())
    gathered = store.gather(torch.arange(12), DEV)
    plain = b["image"].to(DEV)
    res = []
    for img in (gathered, plain):
        nmn.zero_grad(set_to_none=True)
        out = nmn(img, b["program"], b["answer"].to(DEV))
        out["loss"].mean().backward()
        res.append((out["loss"].detach().clone(), out["predictions"].clone(),
                    {n: p.grad.detach().clone() for n, p in nmn.named_parameters()}))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for n in res[0][2]:
        a, c = res[0][2][n], res[1][2][n]
        assert float((a - c).abs().max()) <= 2e-5 * float(c.abs().max()) + 1e-12, n  # (fp32 atomics order only)


@pytest.mark.parametrize("method", ["dma", "kernel"])
def test_prefetching_loader_yields_every_batch_with_its_features(method):
    """Both ingest paths: the copy engines into an NCHW batch (default) and the PCIe-reading gather kernel (NHWC)."""
    from probnmn.data.feature_store import PinnedFeatureStore, PrefetchingLoader

    rng = np.random.Generator(np.random.Philox(1))
    feats = rng.standard_normal((50, 1024, 14, 14)).astype(np.float32)
    store = PinnedFeatureStore(feats)
    host_batches = []
    for k in range(5):
        idx = torch.from_numpy(rng.integers(0, 50, 8 if k != 3 else 5))
        host_batches.append({"image_index": idx, "question": torch.full((idx.numel(), 4), k), "answer": idx % 28,
                             "supervision": (idx % 2)})
    seen = 0
    for k, batch in enumerate(PrefetchingLoader(host_batches, store, DEV, method=method)):
        hb = host_batches[k]
        assert set(batch) == {"image", "question", "answer", "supervision"}
        assert batch["supervision"].device.type == "cpu" and batch["question"].is_cuda
        assert batch["image"].is_contiguous() == (method == "dma")
        # consume on the compute stream (a kernel that reads the whole batch), then check
        total = batch["image"].double().sum()
        assert torch.equal(batch["image"].cpu(), torch.from_numpy(feats[hb["image_index"].numpy()]))
        assert abs(float(total) - float(feats[hb["image_index"].numpy()].astype(np.float64).sum())) < 1e-3 * abs(float(total)) + 1e-2
        assert torch.equal(batch["question"].cpu(), hb["question"])
        seen += 1
    assert seen == 5


def test_copy_rows_rejects_an_index_outside_the_store():
    from probnmn.data.feature_store import PinnedFeatureStore

    store = PinnedFeatureStore(np.zeros((4, 8, 2, 2), np.float32))
    with pytest.raises(IndexError):
        store.copy_rows(torch.tensor([0, 4]), DEV)
    got = store.copy_rows(torch.tensor([3, 0, 3]), DEV)
    assert tuple(got.shape) == (3, 8, 2, 2) and got.is_contiguous()


def test_resident_store_feeds_the_network_without_a_copy():
    """DeviceFeatureStore: all rows in HBM in the stem's NHWC layout; a batch is (store, host indices) and the stem conv1 /
    stem weight-gradient records point straight at the selected rows.  Exactness: the resident rows equal host indexing
    bit for bit; a module-training step and a joint-training step fed with ``store.batch(idx)`` give the SAME losses
    (bit-equal forward) and gradients (up to the order of the weight gradients' atomic adds) as the steps fed with the
    gathered tensor -- with repeated and out-of-order indices, and a subset taken by the joint step."""
    from probnmn.data.feature_store import DeviceFeatureStore, ResidentRows
    from probnmn.data.synthetic import synthetic_batch
    from probnmn.models import NeuralModuleNetwork, ProgramGenerator, ProgramPrior, QuestionReconstructor
    from probnmn.trainers.joint_training import JointTrainingStep
    from probnmn.trainers.module_training import ModuleTrainingStep
    from probnmn.vocabulary import Vocabulary

    dev = torch.device("cuda:0")
    vocab = Vocabulary.clevr()
    g = torch.Generator().manual_seed(5)
    feats = torch.relu(torch.randn(23, 1024, 14, 14, generator=g))
    store = DeviceFeatureStore(feats.numpy(), dev, chunk_rows=7)  # (several chunks, a ragged last one)
    idx = torch.tensor([3, 22, 3, 0, 17, 9, 9, 21, 1, 14, 6, 2])
    rows = store.batch(idx)
    assert isinstance(rows, ResidentRows) and rows.shape == (12, 1024, 14, 14)
    assert torch.equal(rows.materialize().cpu(), feats[idx])
    with pytest.raises(IndexError):
        store.batch(torch.tensor([0, 23]))

    batch = synthetic_batch(vocab, 12, seed=8)
    batch["image"] = feats[idx]

    def module_step(image):
        torch.manual_seed(0)
        net = NeuralModuleNetwork(vocab, class_projection_channels=128, classifier_linear_size=64).to(dev)
        step = ModuleTrainingStep(net, lr=1e-4, report_metrics=False)
        b = {k: (v.to(dev) if k != "program" else v) for k, v in batch.items() if k != "image"}
        b["image"] = image
        out = step.step(b)
        torch.cuda.synchronize()
        return out["loss"].detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}

    loss_t, grads_t = module_step(batch["image"].to(dev))
    loss_r, grads_r = module_step(rows)
    assert torch.equal(loss_t, loss_r)
    for k in grads_t:
        assert float((grads_t[k] - grads_r[k]).abs().max()) <= 1e-4 * (float(grads_t[k].abs().max()) + 1e-12), k

    def joint_step(image):
        torch.manual_seed(1)
        pg, qr = ProgramGenerator(vocab).to(dev), QuestionReconstructor(vocab).to(dev)
        prior = ProgramPrior(vocab, hidden_size=256).to(dev)
        net = NeuralModuleNetwork(vocab, class_projection_channels=128, classifier_linear_size=64).to(dev)
        step = JointTrainingStep(pg, qr, prior, net, objective="ours", alpha=100.0, beta=0.1, gamma=1.0, delta=0.99, lr=1e-4)
        b = {k: v.to(dev) for k, v in batch.items() if k != "image"}
        b["supervision"] = batch["supervision"]
        b["image"] = image
        out = step.step(b)
        torch.cuda.synchronize()
        return float(out["objective"]), out["programs"].cpu(), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}

    obj_t, z_t, g_t = joint_step(batch["image"].to(dev))
    obj_r, z_r, g_r = joint_step(rows)
    assert torch.equal(z_t, z_r) and obj_t == obj_r
    for k in g_t:
        assert float((g_t[k] - g_r[k]).abs().max()) <= 1e-4 * (float(g_t[k].abs().max()) + 1e-12), k


@pytest.mark.gpu
def test_resident_store_adopts_device_features_in_place():
    """DeviceFeatureStore.from_device: the feature extractor's channels_last output becomes the store without a copy; rows
    read through it equal the tensor's."""
    from probnmn.data.feature_store import DeviceFeatureStore

    dev = torch.device("cuda:0")
    feats = torch.randn(12, 1024, 14, 14, device=dev).contiguous(memory_format=torch.channels_last)
    store = DeviceFeatureStore.from_device(feats)
    assert len(store) == 12 and store.data_ptr() == feats.data_ptr() and store.image_feature_size == (1024, 14, 14)
    rows = store.batch([7, 0, 7, 11])
    assert rows.shape == (4, 1024, 14, 14)
    assert torch.equal(rows.materialize(), feats[[7, 0, 7, 11]])
    assert rows.pointers().tolist() == [feats.data_ptr() + i * 1024 * 196 * 4 for i in (7, 0, 7, 11)]
    with pytest.raises(ValueError):
        DeviceFeatureStore.from_device(torch.randn(2, 1024, 14, 14, device=dev))  # NCHW storage
