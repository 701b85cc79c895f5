from .bc import BC, BCTrainer
from .bcql import BCQL, BCQLTrainer
from .cpq import CPQ, CPQTrainer

__all__ = ["BC", "BCTrainer", "BCQL", "BCQLTrainer", "CPQ", "CPQTrainer"]
