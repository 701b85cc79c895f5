from .bc import BC, BCTrainer
from .bcql import BCQL, BCQLTrainer
from .cdt import CDT, CDTTrainer
from .cpq import CPQ, CPQTrainer

__all__ = ["BC", "BCTrainer", "BCQL", "BCQLTrainer", "CDT", "CDTTrainer", "CPQ", "CPQTrainer"]
