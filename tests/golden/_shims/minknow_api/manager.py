"""Import shim (fixture generation only)."""


class Manager:
    pass


class FlowCellPosition:
    pass
