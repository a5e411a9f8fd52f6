"""Summarise rocprofv3 --pmc result dbs: per kernel name, per counter: mean value per dispatch."""
import sqlite3, sys, collections

def summarize(db_path, name_like='%'):
    db = sqlite3.connect(db_path); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    rows = cur.execute("select * from counters_collection limit 1").fetchall()
    cols = [r[1] for r in cur.execute("pragma table_info('counters_collection')")]
    q = cur.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection where kernel_name like ? group by kernel_name, counter_name", (name_like,)).fetchall()
    return cols, q

if __name__ == '__main__':
    for p in sys.argv[1:]:
        try:
            cols, q = summarize(p)
        except Exception as e:
            print(p, 'ERR', e); continue
        print('#', p)
        for k, c, n, avg, tot in q:
            print("%-50s %-28s n=%-5d avg=%-16.1f" % (k[:50], c, n, avg))
