__all__ = ["Connector", "ConnectorParams", "DiscreteConnector", "StateSpace", "EuclideanSpace", "PlannerParams", "PRM", "PRMStar", "RRT", "RRTStar",
           "Node", "knn_prefix"]

from .connectors import Connector, ConnectorParams, DiscreteConnector
from .roadmap import StateSpace, EuclideanSpace, PlannerParams, PRM, PRMStar, RRT, RRTStar, Node, knn_prefix
