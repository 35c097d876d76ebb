__all__ = ["Connector", "ConnectorParams", "DiscreteConnector"]

from .connectors import Connector, ConnectorParams, DiscreteConnector
