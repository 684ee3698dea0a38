"""Parameter-arena layout of the native HMR engine and the conversions to / from the reference's
``state_dict`` (names and shapes of reference model/hmr.py:63-106, i.e. torchvision-ResNet names +
fc1/fc2/decpose/decshape/deccam + the three ``init_*`` buffers).

The engine keeps all 169 tensors in one flat fp32 arena in kernel-friendly layouts:
  conv   (Cout,Cin,R,S)  ->  [R][S][Cin_pad][Cout]         (Cin 3 -> 4 for the stem)
  fc1    (1024,2205)     ->  [1024][2208]                   (row stride multiple of 4)
  decpose/decshape/deccam -> one [160][1024] matrix + [160] bias (rows 157..159 zero)
so Adam / fast-weight / EMA are single elementwise launches.  Pure host-side bookkeeping."""
from __future__ import annotations

import ctypes
from typing import Dict, List

import torch

K_CONV_W, K_NORM_W, K_NORM_B, K_FC_W, K_FC_B, K_DEC_W, K_DEC_B = range(7)
STATE_LD = 160
DEC_ROWS = (("decpose", 0, 144), ("decshape", 144, 10), ("deccam", 154, 3))


class HmrLayout:
    def __init__(self, lib, batch: int, height: int = 224, width: int = 224):
        self.lib, self.B, self.H, self.W = lib, batch, height, width
        plan = ctypes.c_void_p()
        rc = lib.dyb_hmr_plan_create(batch, height, width, ctypes.byref(plan))
        if rc != 0:
            raise RuntimeError(f"dyb_hmr_plan_create failed ({rc})")
        self.plan = plan
        self.n_params = int(lib.dyb_hmr_param_floats(plan))
        self.act_floats = int(lib.dyb_hmr_act_floats(plan))
        self.ws_bytes = int(lib.dyb_hmr_workspace_bytes(plan))
        self.tensors: List[dict] = []
        name = ctypes.create_string_buffer(128)
        kind, cpad, off = ctypes.c_int(), ctypes.c_int(), ctypes.c_longlong()
        dims = (ctypes.c_int * 4)()
        for i in range(lib.dyb_hmr_num_tensors(plan)):
            lib.dyb_hmr_tensor_info(plan, i, ctypes.cast(name, ctypes.c_void_p), 128, ctypes.cast(ctypes.pointer(kind), ctypes.c_void_p),
                                    ctypes.cast(ctypes.pointer(off), ctypes.c_void_p), ctypes.cast(dims, ctypes.c_void_p),
                                    ctypes.cast(ctypes.pointer(cpad), ctypes.c_void_p))
            self.tensors.append(dict(name=name.value.decode(), kind=kind.value, offset=off.value,
                                     dims=list(dims), cin_pad=cpad.value))
        self.features: List[dict] = []
        rs = ctypes.c_int()
        for w in range(15):
            lib.dyb_hmr_feature_info(plan, w, ctypes.cast(ctypes.pointer(off), ctypes.c_void_p),
                                     ctypes.cast(dims, ctypes.c_void_p), ctypes.cast(ctypes.pointer(rs), ctypes.c_void_p))
            self.features.append(dict(offset=off.value, dims=list(dims), row_stride=rs.value))
        # train mode: features 7+3t are the hidden vectors AFTER drop1 (reference model/hmr.py:165-166)
        self.features_train: List[dict] = []
        for w in range(15):
            lib.dyb_hmr_feature_info_ex(plan, w, 1, ctypes.cast(ctypes.pointer(off), ctypes.c_void_p),
                                        ctypes.cast(dims, ctypes.c_void_p), ctypes.cast(ctypes.pointer(rs), ctypes.c_void_p))
            self.features_train.append(dict(offset=off.value, dims=list(dims), row_stride=rs.value))
        # whole-call hipGraph caching inside the engine.  OFF by default: measured on MI355X / ROCm 7.2 with a
        # 97 % replay rate the host needs MORE time per frame (19.1 ms vs 15.2 ms eager) - hipGraphLaunch of a
        # 180-330-node graph is slower than the engine's own C++ launch loop.  DYB_GRAPHS=1 enables it.
        import os
        self.graphs = os.environ.get("DYB_GRAPHS", "0") == "1"
        lib.dyb_hmr_set_graph_mode(plan, 1 if self.graphs else 0)
        self.bf16 = False
        self.off_rotmat = int(lib.dyb_hmr_act_offset_rotmat(plan))
        self.off_state = int(lib.dyb_hmr_act_offset_state(plan))

    def __del__(self):
        try:
            self.lib.dyb_hmr_plan_destroy(self.plan)
        except Exception:
            pass

    def set_bf16(self, on: bool):
        """bf16 matrix-core variant of this plan's convolutions (fp32 master weights / activations / accumulators; operand
        tiles rounded as they are staged).  Never the parity default."""
        self.bf16 = bool(on)
        rc = self.lib.dyb_hmr_set_bf16(self.plan, 1 if on else 0)
        if rc != 0:
            raise RuntimeError(f"dyb_hmr_set_bf16 failed ({rc})")

    def graph_stats(self):
        st = (ctypes.c_longlong * 10)()
        self.lib.dyb_hmr_graph_stats(self.plan, ctypes.cast(st, ctypes.c_void_p))
        return dict(replays=int(st[0]), eager=int(st[1]), captures=int(st[2]), fwd_keys=int(st[3]), bwd_keys=int(st[4]),
                    fail=dict(begin=int(st[5]), body=int(st[6]), end=int(st[7]), instantiate=int(st[8]), launch=int(st[9])))

    # ---- sizes -------------------------------------------------------------------------------
    def numel(self, t: dict) -> int:
        d, k = t["dims"], t["kind"]
        if k == K_CONV_W:
            return d[2] * d[3] * t["cin_pad"] * d[0]
        if k in (K_FC_W, K_DEC_W):
            return d[0] * d[2]
        return d[0]

    # ---- reference state_dict -> arena ---------------------------------------------------------
    def pack(self, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
        """`sd` uses un-prefixed reference names.  Returns a CPU fp32 arena (pad gaps zero)."""
        flat = torch.zeros(self.n_params, dtype=torch.float32)
        for t in self.tensors:
            k, d, o = t["kind"], t["dims"], t["offset"]
            if k == K_CONV_W:
                w = sd[t["name"]].float()
                assert list(w.shape) == [d[0], d[1], d[2], d[3]], (t["name"], w.shape)
                w = w.permute(2, 3, 1, 0)                                  # R,S,Cin,Cout
                if t["cin_pad"] != d[1]:
                    w = torch.cat([w, torch.zeros(d[2], d[3], t["cin_pad"] - d[1], d[0])], 2)
                v = w.reshape(-1)
            elif k == K_FC_W:
                w = sd[t["name"]].float()
                assert list(w.shape) == [d[0], d[1]], (t["name"], w.shape)
                v = torch.zeros(d[0], d[2])
                v[:, :d[1]] = w
                v = v.reshape(-1)
            elif k == K_DEC_W:
                v = torch.zeros(d[0], d[2])
                for nm, r0, n in DEC_ROWS:
                    v[r0:r0 + n] = sd[nm + ".weight"].float()
                v = v.reshape(-1)
            elif k == K_DEC_B:
                v = torch.zeros(d[0])
                for nm, r0, n in DEC_ROWS:
                    v[r0:r0 + n] = sd[nm + ".bias"].float()
            else:
                v = sd[t["name"]].float().reshape(-1)
                assert v.numel() == d[0], t["name"]
            flat[o:o + v.numel()] = v
        return flat

    # ---- arena -> reference-named tensors (weights or gradients) -------------------------------
    def unpack(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        flat = flat.detach().cpu()
        out: Dict[str, torch.Tensor] = {}
        for t in self.tensors:
            k, d, o = t["kind"], t["dims"], t["offset"]
            v = flat[o:o + self.numel(t)]
            if k == K_CONV_W:
                w = v.view(d[2], d[3], t["cin_pad"], d[0])[:, :, :d[1], :]
                out[t["name"]] = w.permute(3, 2, 0, 1).contiguous()
            elif k == K_FC_W:
                out[t["name"]] = v.view(d[0], d[2])[:, :d[1]].contiguous()
            elif k == K_DEC_W:
                m = v.view(d[0], d[2])
                for nm, r0, n in DEC_ROWS:
                    out[nm + ".weight"] = m[r0:r0 + n].contiguous()
            elif k == K_DEC_B:
                for nm, r0, n in DEC_ROWS:
                    out[nm + ".bias"] = v[r0:r0 + n].contiguous()
            else:
                out[t["name"]] = v.clone()
        return out

    @staticmethod
    def init_state(sd: Dict[str, torch.Tensor]) -> torch.Tensor:
        """[1][160] = init_pose | init_shape | init_cam | 0 (reference model/hmr.py:100-106)."""
        st = torch.zeros(1, STATE_LD)
        st[0, :144] = sd["init_pose"].reshape(-1).float()
        st[0, 144:154] = sd["init_shape"].reshape(-1).float()
        st[0, 154:157] = sd["init_cam"].reshape(-1).float()
        return st
