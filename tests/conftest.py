import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
os.environ.setdefault("MIOPEN_LOG_LEVEL", "3")   # errors only: immediate-mode workspace warnings are noise


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs the read-only reference tree at /root/reference")


def pytest_sessionstart(session):
    # the shipped MIOpen find-db / kernel cache (explicit since round 3: importing the package no longer sets it up)
    import structure_knowledge_distillation_amd as S
    S.configure_miopen()


def pytest_collection_modifyitems(config, items):
    import torch
    has_gpu = torch.cuda.is_available()
    skip_gpu = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords and not has_gpu:
            item.add_marker(skip_gpu)
