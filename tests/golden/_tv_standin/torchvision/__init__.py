"""Offline, topology-only stand-in for the `torchvision` package (not installed in this image).

Used ONLY by tests/golden/make_golden.py in the development container so that the
upstream `mmvit4.py` can be imported as an oracle.  The upstream
encoder reads nothing but layer topology from `resnet50()` and overwrites every copied
weight afterwards, so no pretrained data is needed or fetched.
"""
