"""pytest config: registers the ``gpu`` marker and puts the product package + repo root on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "speech-enhancement-pytorch_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A bound on every test (pytest-timeout, when the plugin is there): a wedged GPU kernel or an oversubscribed CPU oracle
    then fails ONE test after 10 minutes instead of holding the whole run."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    import pytest
    for it in items:
        if it.get_closest_marker("timeout") is None:
            it.add_marker(pytest.mark.timeout(600))
