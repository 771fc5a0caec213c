"""hipops.state_key: the fingerprint that decides when packed weights / BN plans are rebuilt.  It caches the list of registered
tensors (walking the module tree costs more than a forward's other host work) and must still see every kind of change."""
import torch

from msnets_amd import hipops
from msnets_amd.gcnet_3dcnn import GCNet_CostVolumeAggre


def test_state_key_sees_every_tracked_change():
    m = GCNet_CostVolumeAggre(32).eval()
    k = hipops.state_key(m)
    assert hipops.state_key(m) == k                                  # stable while nothing changes
    with torch.no_grad():
        m.conv3dbn_1[0].weight.mul_(2.0)                             # in-place edit: version counter
    k1 = hipops.state_key(m)
    assert k1 != k
    m.conv3dbn_1[1].running_var = torch.ones(32)                     # a re-assigned buffer is a new object in the owner's dict
    k2 = hipops.state_key(m)
    assert k2 != k1
    m.deconv5.bias = torch.nn.Parameter(torch.zeros(1))              # a re-assigned parameter
    k3 = hipops.state_key(m)
    assert k3 != k2
    m.load_state_dict({n: v.clone() for n, v in m.state_dict().items()})      # copy_ into every tensor
    k4 = hipops.state_key(m)
    assert k4 != k3
    m.double()                                                       # new storage for every parameter, new buffer objects
    assert hipops.state_key(m) != k4
    k5 = hipops.state_key(m)
    m.invalidate_plans()                                             # re-walks the tree; same tensors, same key
    assert hipops.state_key(m) == k5
