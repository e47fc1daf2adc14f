"""recipe for pinning the oracle against the real reference on a machine with cargo (README.md) -- TEST INFRASTRUCTURE"""
