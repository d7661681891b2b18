"""Python livelink client: the producer side of the engine's TCP scene link.

Mirror of Engine/ZeldaPython/ZeldaUntitled.py:12-26 (`sendDataToEngine`): connect to localhost:<port>, `sendall` the
whole JSON document in one go (the server does exactly ONE recv of <= 65720 bytes per connection, ZE:972-973,1683),
then `recv(1024)`, which returns b'' because the server never sends a payload and half-closes (ZE:1699).
"""
import json
import socket

RECV_MAX = 65720      # ZE:972-973
DEFAULT_PORT = 8080   # ZE:1636


def sendDataToEngine(data, port=DEFAULT_PORT, host="localhost", timeout=5.0):
    """Same name, arguments and behaviour as the reference client; returns the bytes received (b'' from the engine)."""
    payload = data.encode() if isinstance(data, str) else bytes(data)
    if len(payload) > RECV_MAX:
        raise ValueError("scene JSON is %d bytes; the engine reads at most %d in its single recv" % (len(payload), RECV_MAX))
    try:
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.settimeout(timeout)
            s.connect((host, port))
            s.sendall(payload)
            response = s.recv(1024)
            print("Received:", response.decode())
            return response
    except ConnectionRefusedError:
        print(f"Connection to port {port} failed. Make sure there's a server listening on this port.")
    except Exception as e:      # noqa: BLE001  (the reference client swallows and prints, ZeldaUntitled.py:25-26)
        print(f"An error occurred: {e}")
    return None


def send_world(world, port=DEFAULT_PORT, host="localhost"):
    """json.dumps with CPython's default separators, exactly what ZeldaUntitled.py:163 would put on the wire."""
    return sendDataToEngine(json.dumps(world), port, host)
