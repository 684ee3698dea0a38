"""The loaders for the reference's real on-disk formats, exercised on synthetic content written in
exactly those formats: SMPL_*.pkl as smplx consumes it (chumpy-free dict with a scipy-sparse J_regressor, a
kintree_table whose root parent is 2**32-1, 300-wide shapedirs), J_regressor_{extra,h36m}.npy, smpl_mean_params.npz,
gmm_08.pkl (float64 means / covars / weights), basemodel.pt ({'model': state_dict with the MAML wrapper's 'module.'
prefix}) - then a BaseAdaptor built from that directory tree the way the reference builds it (no bundle), running on the
kernel emulator."""
import ctypes
import os
import pickle

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import rel_err


@pytest.fixture(scope="module")
def emu_lib():
    from emu.build_emu import build
    from dynaboa_amd import _abi, _lib
    lib = _abi.bind(ctypes.CDLL(build()))
    saved = _lib._lib
    _lib.use_library(lib)
    yield lib
    _lib._lib = saved


def write_smpl_pkl(path, tab, rng):
    V = tab["v_template"].shape[0]
    sd300 = np.concatenate([tab["shapedirs"].astype(np.float64), rng.normal(0, 0.01, (V, 3, 290))], 2)     # real files carry 300 betas
    kin = np.stack([np.asarray(tab["parents"]).astype(np.int64) % (2 ** 32), np.arange(24)]).astype(np.uint32)
    d = dict(v_template=tab["v_template"].astype(np.float64), shapedirs=sd300,
             posedirs=tab["posedirs"].T.reshape(V, 3, 207).astype(np.float64),
             J_regressor=sp.csc_matrix(tab["J_regressor"].astype(np.float64)), weights=tab["lbs_weights"].astype(np.float64),
             kintree_table=kin, f=tab["faces"].astype(np.uint32), bs_type="lrotmin", bs_style="lbs")
    with open(path, "wb") as f:
        pickle.dump(d, f, protocol=2)


@pytest.fixture(scope="module")
def data_tree(tmp_path_factory):
    from dynaboa_amd import assets
    root = tmp_path_factory.mktemp("refdata")
    data = root / "data"
    (data / "smpl").mkdir(parents=True)
    (data / "spin_data").mkdir()
    rng = np.random.default_rng(5)
    tabs = {g: assets.make_synthetic_smpl(i) for i, g in enumerate(("NEUTRAL", "MALE", "FEMALE"))}
    for g, t in tabs.items():
        write_smpl_pkl(str(data / "smpl" / f"SMPL_{g}.pkl"), t, rng)
    np.save(data / "J_regressor_extra.npy", tabs["NEUTRAL"]["J_regressor_extra"])
    np.save(data / "J_regressor_h36m.npy", tabs["NEUTRAL"]["J_regressor_h36m"])
    mp = assets.make_smpl_mean_params(identity_pose=False, seed=3)
    np.savez(data / "smpl_mean_params.npz", **mp)
    ck = assets.make_synthetic_checkpoint(22, mp, randomize_norm=True, prefix="module.")
    torch.save(ck, data / "basemodel.pt")
    covs = []
    for _ in range(8):
        a = rng.normal(0, 1, (69, 69))
        covs.append(a @ a.T / 69 + 0.5 * np.eye(69))
    gmm = dict(means=rng.normal(0, 0.3, (8, 69)), covars=np.stack(covs), weights=np.full(8, 0.125))
    with open(data / "spin_data" / "gmm_08.pkl", "wb") as f:
        pickle.dump(gmm, f, protocol=2)
    return dict(root=root, tabs=tabs, ck=ck, mp=mp, gmm=gmm)


def test_smpl_pkl_roundtrip(data_tree):
    from dynaboa_amd import assets
    d = data_tree["root"] / "data"
    t = assets.load_smpl_pkl(str(d / "smpl" / "SMPL_MALE.pkl"), str(d / "J_regressor_extra.npy"), str(d / "J_regressor_h36m.npy"))
    ref = data_tree["tabs"]["MALE"]
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights"):
        assert t[k].dtype == np.float32 and t[k].shape == ref[k].shape, k
        assert rel_err(t[k], ref[k]) < 1e-6, k
    assert list(t["parents"]) == list(ref["parents"])                    # root parent 2**32-1 -> -1
    assert np.array_equal(t["faces"], ref["faces"])
    # the extra / h36m regressors come from the .npy files (the NEUTRAL ones were written)
    assert np.array_equal(t["J_regressor_extra"], data_tree["tabs"]["NEUTRAL"]["J_regressor_extra"])


def test_gmm_pickle_buffers(data_tree):
    from dynaboa_amd import assets
    g = data_tree["gmm"]
    b = assets.gmm_buffers_from_pickle(str(data_tree["root"] / "data" / "spin_data" / "gmm_08.pkl"))
    assert b["means"].dtype == np.float32 and b["precisions"].shape == (8, 69, 69) and b["nll_weights"].shape == (1, 8)
    for m in range(8):          # precision really is the inverse covariance
        assert np.abs(b["precisions"][m].astype(np.float64) @ g["covars"][m] - np.eye(69)).max() < 5e-3
    sq = np.sqrt(np.linalg.det(g["covars"]))
    assert rel_err(b["nll_weights"].ravel(), g["weights"] / ((2 * np.pi) ** 34.5 * sq / sq.min())) < 1e-5


def test_adaptor_from_reference_tree(emu_lib, data_tree, monkeypatch):
    """BaseAdaptor with NO bundle reads the reference's relative paths (base_adaptor.py:116-125,144-149;
    model/hmr.py:100-103): weights equal the checkpoint, SMPL models are the three gendered files, the prior is the
    pickle's."""
    _adaptor_from_reference_tree("cpu", data_tree, monkeypatch)


@pytest.mark.gpu
def test_adaptor_from_reference_tree_gpu(data_tree, monkeypatch):
    """The same loaders feeding the real library on cuda:0 (the asset-format tests had no GPU run)."""
    _adaptor_from_reference_tree("cuda:0", data_tree, monkeypatch)


def _adaptor_from_reference_tree(device, data_tree, monkeypatch):
    from dynaboa_amd import benchmark as DB
    monkeypatch.chdir(data_tree["root"])
    o = DB.frame_only_options(inner_step=1)
    o.model_file = "data/basemodel.pt"
    ad = DB.Adaptor(o, None, device=device)
    sd = ad.model.state_dict()
    ck = data_tree["ck"]["model"]
    assert set(sd) == set(ck)
    for k in ("module.conv1.weight", "module.layer3.2.bn2.bias", "module.fc1.weight", "module.decshape.bias", "module.init_pose"):
        assert torch.equal(sd[k].cpu(), ck[k]), k
    tab = data_tree["tabs"]
    assert rel_err(ad.smpl_female.v_template.cpu().numpy(), tab["FEMALE"]["v_template"]) < 1e-6
    assert rel_err(ad.smpl_male.posedirs.cpu().numpy(), tab["MALE"]["posedirs"]) < 1e-6
    assert tuple(ad.J_regressor.shape) == (17, 6890)
    assert rel_err(ad.gmm_f.means.cpu().numpy(), data_tree["gmm"]["means"].astype(np.float32)) < 1e-6
    # and the loaded SMPL runs: one LBS forward on the emulated kernels against the oracle on the same tables
    from oracle import ref_cpu as O
    g = torch.Generator().manual_seed(3)
    betas = torch.randn(1, 10, generator=g) * 0.5
    rot = O.smplx_rodrigues(torch.randn(24, 3, generator=g) * 0.3).view(1, 24, 3, 3)
    out = ad.smpl_neutral(betas=betas.to(device), body_pose=rot[:, 1:].to(device), global_orient=rot[:, :1].to(device), pose2rot=False)
    T = O.smpl_tables_to_torch(tab["NEUTRAL"])
    verts, j49 = O.smpl_forward(T, betas, rot[:, 1:], rot[:, :1], pose2rot=False)
    assert rel_err(out.vertices.cpu().numpy(), verts.numpy()) < 1e-5
    assert rel_err(out.joints.cpu().numpy(), j49.numpy()) < 1e-5


@pytest.mark.slow
def test_adaptor_real_data_path_one_frame(emu_lib, data_tree, monkeypatch):
    """The reference's CLI path end to end with NO synthetic bundle: Adaptor(options) builds the 3DPW loader and the exemplar
    set from a reference-style data tree (default flags: retrieval=1, labelled exemplars, teacher, dynamic loop), excute()
    walks the first frame - decode, device crop, lower level with a retrieved exemplar, upper level, Adam, metrics."""
    _real_data_path("cpu", 1, data_tree, monkeypatch)


@pytest.mark.gpu
def test_adaptor_real_data_path_gpu(data_tree, monkeypatch):
    """The same on cuda:0 over four frames (the motion term's history pair exists from frame 3 on with interval 2): the real
    cluster-argmin retrieval callback of the native stepper, device crop preprocessing, default flags."""
    _real_data_path("cuda:0", 4, data_tree, monkeypatch)


def _real_data_path(device, nframes, data_tree, monkeypatch):
    from test_preprocess import populate_stream
    from dynaboa_amd import benchmark as DB
    tree = populate_stream(data_tree["root"])
    monkeypatch.chdir(data_tree["root"])
    o = DB.parser.parse_args([])
    o.model_file, o.pw3d_root, o.h36m_root = "data/basemodel.pt", tree["imgroot"], tree["h36root"]
    o.expdir = str(data_tree["root"] / "exps")
    if nframes > 1:
        o.interval = 2
    ad = DB.Adaptor(o, None, device=device)
    assert len(ad.dataloader) == 9 and len(ad.h36m_dataset) == 12 and tuple(ad.centers.shape) == (3, 2048)
    it = iter(ad.dataloader)
    frames = [next(it) for _ in range(nframes)]
    theta0 = ad.model.module.theta.detach().clone()
    res = ad.excute(frames, nframes=nframes)
    assert np.isfinite(res["mpjpe"][0]).all() and np.isfinite(res["pampjpe"][0]).all() and len(res["mpjpe"]) == nframes
    assert float((ad.model.module.theta.detach() - theta0).abs().max()) > 0
    assert "ll/labled_loss" in ad.last_summaries and "teacher/loss" in ad.last_summaries and len(ad.optim_step_record) == nframes
    if nframes > 3:
        assert "ul/motion_loss" in ad.last_summaries
    assert os.path.exists(os.path.join(o.expdir, o.expname, "seq_order.record"))
