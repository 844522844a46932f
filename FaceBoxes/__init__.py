"""Drop-in for the reference's `FaceBoxes` package (`from FaceBoxes import FaceBoxes`, synergy3DMM.py:6): the detector class
served by the HIP kernels of synergynet_amd.  Put the reference's weights at FaceBoxes/weights/FaceBoxesProd.pth."""
from synergynet_amd.faceboxes import FaceBoxes  # noqa: F401
