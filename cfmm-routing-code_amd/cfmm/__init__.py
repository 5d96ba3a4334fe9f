"""cfmm -- MI355X-native optimal routing for constant function market makers.

Drop-in for the `cp.Problem(obj, cons).solve()` call of angeris/cfmm-routing-code
(arbitrage.py:81-82, liquidation.py:84-85, two-asset.py:90-91); see problem.Problem.
"""
from .problem import Problem, Utility, Arbitrage, Liquidate, Swap, LogUtility, QuadraticUtility, HostComm, pack, shard_network, start_prices
from ._lib import CfmmError, GE, EQ, FREE, ULOG, UQUAD
from . import distributed
from . import cvx

__all__ = ["Problem", "Utility", "Arbitrage", "Liquidate", "Swap", "pack", "shard_network",
           "start_prices", "HostComm", "CfmmError", "GE", "EQ", "FREE", "ULOG", "UQUAD", "LogUtility", "QuadraticUtility", "distributed", "cvx"]
