"""MI355X-native batched kinematics + collision validity behind numbotics' Arm / GraphChain /
DiscreteConnector signatures.  See DESIGN.md."""
__version__ = "0.1.0"
