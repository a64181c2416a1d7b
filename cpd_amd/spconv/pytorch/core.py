import torch

from ... import ops


class SparseConvTensor:
    """[SPCONV] SparseConvTensor(features[N,C], indices[N,4] int32 (b,z,y,x), spatial_shape, batch_size)
    as constructed at spconv_backbone.py:524-529. `indice_dict` caches, per indice_key, the rulebook
    (neighbour table) and site index built by the HIP kernels."""

    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, benchmark=False):
        # `features` may arrive as ops.PairRows (fp16-pair rows left by a fused split-fp16 layer): the next fused layer takes the stored
        # bits as they are; anyone who READS `.features` gets fp32 rows, decoded once and kept (h + l is exact in fp32)
        self._pairs = features if isinstance(features, ops.PairRows) else None
        self._features = None if self._pairs is not None else features
        self.indices = indices.int() if indices.dtype != torch.int32 else indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {}
        self.grid = grid
        self.benchmark = benchmark
        self._site_index = None   # SiteIndex of (indices, spatial_shape), built lazily
        self._row_canon = None    # (canonical site list, old_to_new, chunk) of a level kept in tap-pattern row order (conv.py)

    @property
    def features(self):
        if self._features is None and self._pairs is not None:
            self._features = self._pairs.float_rows()
        return self._features

    @features.setter
    def features(self, value):
        self._pairs = value if isinstance(value, ops.PairRows) else None
        self._features = None if self._pairs is not None else value

    @property
    def spatial_size(self):
        n = 1
        for s in self.spatial_shape:
            n *= s
        return n

    def site_index(self):
        if self._site_index is None:
            self._site_index = ops.SiteIndex.build(self.indices.contiguous(), self.batch_size, self.spatial_shape)
        return self._site_index

    def replace_feature(self, feature):
        """spconv >= 2.1.17 API used by cpd/utils/spconv_utils.py:58-64: shallow copy sharing indices."""
        t = SparseConvTensor(feature, self.indices, self.spatial_shape, self.batch_size, self.grid, self.benchmark)
        t.indice_dict = self.indice_dict
        t._site_index = self._site_index
        t._row_canon = self._row_canon
        return t

    def dense(self, channels_first=True):
        """(B, C, D, H, W) like spconv; height_compression.py:136-138 then views it (B, C*D, H, W)."""
        from ... import autograd_ops
        out = autograd_ops.Densify.apply(self.features, self.indices, self.batch_size, self.spatial_shape)
        return out if channels_first else out.permute(0, 2, 3, 4, 1).contiguous()

    def find_indice_pair(self, key):
        return self.indice_dict.get(key) if key is not None else None
