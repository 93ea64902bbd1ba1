"""A data set resident in HBM: what the device feeds sample from.

Layout (DESIGN.md "data layout"): feature table [N][ldx] fp32 (ldx = F rounded up to 4, so every row
is 16-byte aligned for float4 loads; the last row is the zero pad row appended by Raw_data.pad),
per-row LayerNorm statistics [(N+1)][2] (layer 0 of the DNN normalises raw feature rows, and those
statistics depend on the data only), initial lists [Q][Lmax] int32 (-1 = pad) and labels [Q][Lmax]
fp32.  An MSLR-WEB30K-sized set (3.7 M docs x 136 features) is 2 GB: it fits the 288 GB of one MI355X
hundreds of times over, so every rank keeps a full replica and no feature ever crosses PCIe in the
training loop.
"""
import numpy as np
import torch

from ..ranking_model.DNN import feature_row_stats


class DeviceDataset(object):
    def __init__(self, data_set, device, with_stats=True):
        F = int(data_set.feature_size)
        feats = data_set.features
        if torch.is_tensor(feats):
            host = feats.to(torch.float32)
        else:
            host = torch.from_numpy(np.ascontiguousarray(np.asarray(feats, dtype=np.float32).reshape(-1, F)))
        n = host.shape[0]
        ldx = (F + 3) // 4 * 4
        self.table = torch.zeros((n, ldx), dtype=torch.float32, device=device)
        self.table[:, :F] = host.to(device)
        self.n_rows, self.feature_size, self.ldx = n, F, ldx
        self.stats = feature_row_stats(self.table, F) if with_stats else None
        lists = data_set.initial_list
        if torch.is_tensor(lists) or isinstance(lists, np.ndarray):
            il = torch.as_tensor(lists).to(torch.int32)
            lab = torch.as_tensor(data_set.labels).to(torch.float32)
        else:
            Q = len(lists)
            Lmax = max(len(x) for x in lists)
            il_np = np.full((Q, Lmax), -1, dtype=np.int32)
            lab_np = np.zeros((Q, Lmax), dtype=np.float32)
            for i, (ids, labs) in enumerate(zip(lists, data_set.labels)):
                il_np[i, :len(ids)] = ids
                m = min(len(labs), Lmax)
                lab_np[i, :m] = labs[:m]
            il, lab = torch.from_numpy(il_np), torch.from_numpy(lab_np)
        self.initial_list = il.contiguous().to(device)
        self.labels = lab.contiguous().to(device)
        self.Q, self.Lmax = self.initial_list.shape
        self.key = DeviceDataset._key(data_set)

    @staticmethod
    def _key(data_set):
        return (len(data_set.features), len(data_set.initial_list), int(data_set.rank_list_size))

    @staticmethod
    def of(data_set, device, with_stats=True):
        """Upload once per data set (cached on the object; re-uploaded if pad() changed it)."""
        cached = getattr(data_set, '_ultra_amd_device', None)
        if cached is None or cached.key != DeviceDataset._key(data_set) or cached.table.device != torch.device(device):
            cached = DeviceDataset(data_set, device, with_stats)
            data_set._ultra_amd_device = cached
        return cached

    def lists_batch(self, idx, list_size):
        """docids / labels position-major [list_size][n] for the lists ``idx`` (device int64 tensor)."""
        docs = self.initial_list[idx, :list_size]
        labs = torch.where(docs < 0, torch.zeros((), device=docs.device), self.labels[idx, :list_size])
        docs = torch.where(docs < 0, torch.full((), self.n_rows, dtype=torch.int32, device=docs.device), docs)
        return docs.t().contiguous(), labs.t().contiguous()


class LazyLists(object):
    """[B][L] python lists materialised from a device tensor on first element access; len() is free."""

    def __init__(self, tensor_bl):
        self._t, self._l = tensor_bl, None

    def __len__(self):
        return self._t.shape[0]

    def tolist(self):
        if self._l is None:
            self._l = self._t.cpu().tolist()
        return self._l

    def __getitem__(self, i):
        return self.tolist()[i]

    def __iter__(self):
        return iter(self.tolist())
