"""NoneAttack -- identity attack (reference attack/Black/NoneAttack.py:7-40): the protocol config 1 exercises."""
from .._common import AttackBase


class NoneAttack(AttackBase):
    recommenderModelRequired = False

    def posionDataAttack(self):
        return self.interact
