"""Training path: drop-ins for the reference's utils/network.py *_train networks, train/loss_val.py and
train/trainer.py.  The (1,13) group convolutions run on the HIP library in both directions (forward and data gradient,
csrc/train.hip); everything around them is ordinary PyTorch-ROCm autograd."""
from .network import name2network, PartI_train, PartII_train  # noqa: F401
from .loss_val import name2loss, name2val  # noqa: F401
from .trainer import name2trainer  # noqa: F401
