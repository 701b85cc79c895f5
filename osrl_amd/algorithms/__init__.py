from .bc import BC, BCTrainer
from .bcql import BCQL, BCQLTrainer
from .bearl import BEARL, BEARLTrainer
from .cdt import CDT, CDTTrainer
from .coptidice import COptiDICE, COptiDICETrainer
from .cpq import CPQ, CPQTrainer

__all__ = ["BC", "BCTrainer", "BCQL", "BCQLTrainer", "BEARL", "BEARLTrainer", "CDT", "CDTTrainer", "COptiDICE",
           "COptiDICETrainer", "CPQ", "CPQTrainer"]
