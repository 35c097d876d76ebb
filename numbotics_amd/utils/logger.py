"""Three print wrappers, as in the reference's numbotics/utils/logger.py."""
VERBOSE = True


def info(msg: str):
    if VERBOSE:
        print(f"[NUMBOT INFO] {msg}")


def warning(msg: str):
    if VERBOSE:
        print(f"[NUMBOT WARNING] {msg}")


def error(msg: str):
    print(f"[NUMBOT ERROR] {msg}")
