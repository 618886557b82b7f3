"""Per-kernel parity on a real MI355X: HIP kernels (via the C ABI) vs fp64 restatements."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _checks():
    import gpu_checks
    return gpu_checks.all_checks()


def pytest_generate_tests(metafunc):
    if 'check' in metafunc.fixturenames:
        if torch.cuda.is_available():
            cs = _checks()
            metafunc.parametrize('check', [c[1] for c in cs], ids=[c[0] for c in cs])
        else:
            metafunc.parametrize('check', [None], ids=['needs-gpu'])


def test_kernel(check):
    if check is None:
        pytest.skip('no GPU')
    err, tol = check()
    assert err <= tol, 'error %.3e > tolerance %.1e' % (err, tol)
