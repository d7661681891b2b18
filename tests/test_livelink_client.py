"""Wire behaviour of the Python livelink client against a stand-in server that acts like the engine's listener (CPU only)."""
import json
import socket
import threading

from zeldaengine_amd import livelink, scenes


def _engine_like_server(received):
    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    srv.bind(("127.0.0.1", 0))
    srv.listen(8)

    def run():
        c, _ = srv.accept()
        received.append(c.recv(livelink.RECV_MAX))      # ONE recv (ZE:1683)
        c.shutdown(socket.SHUT_WR)                       # no payload, half close (ZE:1699)
        c.close()
        srv.close()
    t = threading.Thread(target=run, daemon=True)
    t.start()
    return srv.getsockname()[1], t


def test_client_sends_whole_document_and_gets_empty_reply():
    got = []
    port, t = _engine_like_server(got)
    reply = livelink.send_world(scenes.sample_world(), port=port, host="127.0.0.1")
    t.join(5)
    assert reply == b""
    assert got and json.loads(got[0]) == scenes.sample_world()
    assert len(got[0]) == 5727


def test_client_reports_refused_connection_like_the_reference():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    assert livelink.sendDataToEngine("{}", port=port, host="127.0.0.1") is None
