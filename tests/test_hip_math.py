"""GPU: device expf (hanabizero_amd/csrc/hz_common.h) is bit-identical to the host libm expf the reference links
(core/ctree/cnode.cpp:87) on ALL 2^32 float bit patterns."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_expf_all_float_bit_patterns():
    from hanabizero_amd._lib import check, lib
    from oracle.cport import expf_checksums
    out = torch.zeros(4096, dtype=torch.int64, device="cuda")
    check(lib.hz_test_expf_checksum(out.data_ptr(), torch.cuda.current_stream().cuda_stream), "hz_test_expf_checksum")
    dev = out.cpu().numpy().view(np.uint64)
    host = expf_checksums(threads=12)
    bad = np.nonzero(dev != host)[0]
    assert bad.size == 0, "blocks of 2^20 patterns that differ: %s" % bad[:10]


def test_expf_array_spot():
    from hanabizero_amd._lib import check, lib
    from oracle.cport import expf_array
    rng = np.random.RandomState(0)
    x = np.concatenate([rng.randn(100000) * 10, -rng.rand(100000) * 104, [0.0, -0.0, np.inf, -np.inf, np.nan, 88.7, -103.9]]).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty_like(xd)
    check(lib.hz_test_expf(xd.data_ptr(), yd.data_ptr(), x.size, torch.cuda.current_stream().cuda_stream), "hz_test_expf")
    y, ref = yd.cpu().numpy(), expf_array(x)
    same = (y.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(y) & np.isnan(ref))
    assert same.all(), x[~same][:10]
