"""A rendezvous port for the multi-process tests: asked from the kernel (bind to port 0 on 127.0.0.1) instead of derived from the pid - the pid-derived ports of rounds
1 - 4 (29500 ... 37500) sit inside Linux's ephemeral range (32768 - 60999), where any outgoing connection of the box may already hold them (one EADDRINUSE in the round-5 CPU suite)."""
import socket


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])
