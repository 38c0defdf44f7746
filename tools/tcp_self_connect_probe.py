import socket, time, errno
def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p
P = free_port()
print("port", P, "ephemeral range", open("/proc/sys/net/ipv4/ip_local_port_range").read().split())
t0 = time.time(); n = 0
while True:
    n += 1
    c = socket.socket()
    try:
        c.connect(("127.0.0.1", P))
        print(f"attempt {n} after {time.time()-t0:.1f}s: CONNECTED with nobody listening; local {c.getsockname()} peer {c.getpeername()}")
        break
    except ConnectionRefusedError:
        c.close()
    if n >= 400000: print("gave up"); break
srv = socket.socket(); srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
try:
    srv.bind(("127.0.0.1", P)); print("bind succeeded")
except OSError as e:
    print("the rank that is to LISTEN on the port now gets:", errno.errorcode[e.errno])
c.settimeout(2.0)
c.send(b"hello")
print("and the self-connected client reads back its own bytes:", c.recv(16))
