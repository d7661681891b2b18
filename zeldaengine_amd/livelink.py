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
            reply = s.recv(1024)             # the engine half-closes without a payload: b''
            print("livelink: %d bytes delivered to %s:%d, reply %r" % (len(payload), host, port, reply))
            return reply
    except ConnectionRefusedError:
        print("livelink: nobody is listening on %s:%d (is the engine / zr_livelink_serve running?)" % (host, port))
    except OSError as e:        # like the reference client, a failed delivery is reported, not raised (ZeldaUntitled.py:23-26)
        print("livelink: delivery to %s:%d failed: %s" % (host, port, e))
    return None


def send_world(world, port=DEFAULT_PORT, host="localhost"):
    """json.dumps with CPython's default separators, exactly what ZeldaUntitled.py:163 would put on the wire."""
    return sendDataToEngine(json.dumps(world), port, host)
